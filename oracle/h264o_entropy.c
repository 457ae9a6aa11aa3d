/*
 * oracle/h264o_entropy.c -- slice_data() (7.3.4), macroblock_layer() (7.3.5), mb_pred /
 * sub_mb_pred (7.3.5.1/2), residual() (7.3.5.3) with CAVLC (9.2) and CABAC (9.3) parsing,
 * plus the derivations that need neighbour state: Intra4x4/8x8PredMode (8.3.1.1/8.3.2.1) and
 * motion vector prediction (8.4.1).
 *
 * TEST INFRASTRUCTURE ONLY (see h264o.h).
 *
 * Reference counterparts: h264/slice.go:570-830 (NewSliceData, the MB loop -- which stops before
 * residual()), h264/slice.go:252-454 (MbPred), h264/cabac.go:439-553 (arithmetic decoding
 * engine), h264/cabac.go:148-174 (context init), h264/cabac.go:340-428 (Table 9-34 descriptors),
 * h264/cabac.go:557-758 (ctxIdx assignment, neighbour-dependent cases empty there).
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "h264o_int.h"

/* ------------------------------------------------------------------ neighbours */
static h264o_mb *mb_at(h264o_decoder *d, int mbx, int mby) {
    if (mbx < 0 || mby < 0 || mbx >= d->wmb || mby >= d->hmb) return NULL;
    h264o_mb *m = &d->mb[mby * d->wmb + mbx];
    if (m->type == MBT_NONE || m->slice_id != d->slice_id) return NULL;
    return m;
}
#define MB_A(d) mb_at(d, (d)->c.mbx - 1, (d)->c.mby)
#define MB_B(d) mb_at(d, (d)->c.mbx, (d)->c.mby - 1)
#define CUR(d) (&(d)->mb[(d)->c.addr])
/* per-list views of a macroblock's motion (list 0: mv/ref/refid/mvd, list 1: mv1/ref1/refid1/mvd1) */
#define L_MV(m, l) ((l) ? (m)->mv1 : (m)->mv)
#define L_REF(m, l) ((l) ? (m)->ref1 : (m)->ref)
#define L_REFID(m, l) ((l) ? (m)->refid1 : (m)->refid)
#define L_MVD(m, l) ((l) ? (m)->mvd1 : (m)->mvd)
#define IS_SKIP(t) ((t) == MBT_PSKIP || (t) == MBT_BSKIP)

/* luma 4x4 block neighbour at (bx,by) relative to the current MB (bx,by in -1..3):
 * returns the owning MB (NULL if unavailable) and the raster block index inside it. */
static h264o_mb *luma_nb(h264o_decoder *d, int bx, int by, int *ridx) {
    h264o_mb *m;
    if (bx < 0) {
        m = MB_A(d);
        bx += 4;
    } else if (by < 0) {
        m = MB_B(d);
        by += 4;
    } else
        m = CUR(d);
    *ridx = by * 4 + bx;
    return m;
}
static h264o_mb *chroma_nb(h264o_decoder *d, int cx, int cy, int *ridx) {
    h264o_mb *m;
    if (cx < 0) {
        m = MB_A(d);
        cx += 2;
    } else if (cy < 0) {
        m = MB_B(d);
        cy += 2;
    } else
        m = CUR(d);
    *ridx = cy * 2 + cx;
    return m;
}

/* ------------------------------------------------------------------ CAVLC 9.2 */
static int cavlc_coeff_token(h264o_decoder *d, int nC, int *total, int *t1s) {
    h264o_br *b = &d->br;
    uint32_t w = h264o_peek(b, 16);
    if (nC == -1) {
        for (int tc = 0; tc <= 4; tc++)
            for (int t1 = 0; t1 <= (tc < 3 ? tc : 3); t1++) {
                int len = h264o_chroma_dc_token_len[4 * tc + t1];
                if (len && (w >> (16 - len)) == h264o_chroma_dc_token_bits[4 * tc + t1]) {
                    h264o_skip(b, len);
                    *total = tc;
                    *t1s = t1;
                    return 0;
                }
            }
        return -1;
    }
    int tbl = nC < 2 ? 0 : (nC < 4 ? 1 : (nC < 8 ? 2 : 3));
    for (int tc = 0; tc <= 16; tc++)
        for (int t1 = 0; t1 <= (tc < 3 ? tc : 3); t1++) {
            int len = h264o_coeff_token_len[tbl][4 * tc + t1];
            if (len && (w >> (16 - len)) == h264o_coeff_token_bits[tbl][4 * tc + t1]) {
                h264o_skip(b, len);
                *total = tc;
                *t1s = t1;
                return 0;
            }
        }
    return -1;
}

/* 9.2: residual_block_cavlc().  coef[] is filled in scan order [0..maxnum-1]. */
static int cavlc_residual(h264o_decoder *d, int16_t *coef, int maxnum, int nC) {
    h264o_br *b = &d->br;
    int total, t1s;
    memset(coef, 0, sizeof(int16_t) * maxnum);
    if (cavlc_coeff_token(d, nC, &total, &t1s) < 0) return h264o_fail(d, "cavlc: bad coeff_token");
    if (total == 0) return 0;
    if (total > maxnum) return h264o_fail(d, "cavlc: TotalCoeff %d > %d", total, maxnum);
    int level[16], run[16];
    int suffix_len = (total > 10 && t1s < 3) ? 1 : 0;
    for (int i = 0; i < total; i++) {
        if (i < t1s) {
            level[i] = 1 - 2 * (int)h264o_u(b, 1);
            continue;
        }
        int prefix = 0;
        while (h264o_u(b, 1) == 0) {
            prefix++;
            if (prefix > 32 || b->err) return h264o_fail(d, "cavlc: level_prefix overflow");
        }
        int code = (prefix < 15 ? prefix : 15) << suffix_len;
        if (suffix_len > 0 || prefix >= 14) {
            int size = (prefix == 14 && suffix_len == 0) ? 4 : (prefix >= 15 ? prefix - 3 : suffix_len);
            if (size > 0) code += h264o_u(b, size);
        }
        if (prefix >= 15 && suffix_len == 0) code += 15;
        if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
        if (i == t1s && t1s < 3) code += 2;
        level[i] = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
        if (suffix_len == 0) suffix_len = 1;
        if (abs(level[i]) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }
    int zeros_left = 0;
    if (total < maxnum) {
        uint32_t w = h264o_peek(b, 9);
        int found = -1;
        if (maxnum == 4) {
            for (int tz = 0; tz <= 4 - total; tz++) {
                int len = h264o_chroma_dc_total_zeros_len[total - 1][tz];
                if (len && (w >> (9 - len)) == h264o_chroma_dc_total_zeros_bits[total - 1][tz]) {
                    found = tz;
                    h264o_skip(b, len);
                    break;
                }
            }
        } else {
            /* tzVlcIndex = TotalCoeff; for 15-coefficient blocks the same tables apply */
            for (int tz = 0; tz <= 16 - total; tz++) {
                int len = h264o_total_zeros_len[total - 1][tz];
                if (len && (w >> (9 - len)) == h264o_total_zeros_bits[total - 1][tz]) {
                    found = tz;
                    h264o_skip(b, len);
                    break;
                }
            }
        }
        if (found < 0) return h264o_fail(d, "cavlc: bad total_zeros");
        zeros_left = found;
    }
    for (int i = 0; i < total - 1; i++) {
        run[i] = 0;
        if (zeros_left > 0) {
            int t = (zeros_left > 7 ? 7 : zeros_left) - 1;
            uint32_t w = h264o_peek(b, 11);
            int found = -1;
            for (int r = 0; r <= (t == 6 ? 14 : zeros_left); r++) {
                int len = h264o_run_len[t][r];
                if (len && (w >> (11 - len)) == h264o_run_bits[t][r]) {
                    found = r;
                    h264o_skip(b, len);
                    break;
                }
            }
            if (found < 0 || found > zeros_left) return h264o_fail(d, "cavlc: bad run_before");
            run[i] = found;
            zeros_left -= found;
        }
    }
    run[total - 1] = zeros_left;
    int pos = -1;
    for (int i = total - 1; i >= 0; i--) {
        pos += run[i] + 1;
        if (pos >= maxnum) return h264o_fail(d, "cavlc: coefficient position overflow");
        coef[pos] = (int16_t)level[i];
    }
    return total;
}

/* 9.2.1 nC for a luma block at (bx,by) */
static int cavlc_nc_luma(h264o_decoder *d, int bx, int by) {
    int ia, ib;
    h264o_mb *a = luma_nb(d, bx - 1, by, &ia), *bb = luma_nb(d, bx, by - 1, &ib);
    if (a && bb) return (a->nnz[ia] + bb->nnz[ib] + 1) >> 1;
    if (a) return a->nnz[ia];
    if (bb) return bb->nnz[ib];
    return 0;
}
static int cavlc_nc_chroma(h264o_decoder *d, int c, int cx, int cy) {
    int ia, ib;
    h264o_mb *a = chroma_nb(d, cx - 1, cy, &ia), *bb = chroma_nb(d, cx, cy - 1, &ib);
    if (a && bb) return (a->nnz[16 + 4 * c + ia] + bb->nnz[16 + 4 * c + ib] + 1) >> 1;
    if (a) return a->nnz[16 + 4 * c + ia];
    if (bb) return bb->nnz[16 + 4 * c + ib];
    return 0;
}

/* ------------------------------------------------------------------ CABAC engine 9.3.1.2 / 9.3.3.2 */
/* h264/cabac.go:439-446 initDecodingEngine */
void h264o_cabac_init_engine(h264o_decoder *d) {
    d->cod_range = 510;
    d->cod_offset = h264o_u(&d->br, 9);
}

/* 9.3.1.1, (9-5): h264/cabac.go:118-121 PreCtxState, :158-164 state split.  The table set is
 * chosen by slice type / cabac_init_idc (the reference always uses idc 0: Appendix A7). */
static void cabac_init_contexts(h264o_decoder *d) {
    int set = (d->sh.slice_type == 2 || d->sh.slice_type == 4) ? 0 : 1 + d->sh.cabac_init_idc;
    int qp = h264o_clip3(0, 51, d->sh.slice_qp_y);
    for (int i = 0; i < H264O_NCTX; i++) {
        int m = h264o_cabac_mn[set][i][0], n = h264o_cabac_mn[set][i][1];
        int pre = h264o_clip3(1, 126, ((m * qp) >> 4) + n);
        if (pre <= 63)
            d->ctx[i] = (uint8_t)((63 - pre) << 1);
        else
            d->ctx[i] = (uint8_t)(((pre - 64) << 1) | 1);
    }
}

/* 9.3.3.2.1 DecodeDecision + 9.3.3.2.1.1 state transition + 9.3.3.2.2 RenormD.
 * h264/cabac.go:521-540 (BinaryDecision: no transition, no renorm -- Appendix A10),
 * :544-553 (StateTransitionProcess), :503-511 (RenormD). */
int h264o_cabac_decision(h264o_decoder *d, int ctx_idx) {
    uint8_t s = d->ctx[ctx_idx];
    int p = s >> 1, mps = s & 1, bin;
    uint32_t rlps = h264o_range_lps[p][(d->cod_range >> 6) & 3];
    d->cod_range -= rlps;
    if (d->cod_offset >= d->cod_range) {
        bin = !mps;
        d->cod_offset -= d->cod_range;
        d->cod_range = rlps;
        if (p == 0) mps = !mps;
        p = h264o_trans_lps[p];
    } else {
        bin = mps;
        if (p < 62) p++;
    }
    d->ctx[ctx_idx] = (uint8_t)((p << 1) | mps);
    while (d->cod_range < 256) {
        d->cod_range <<= 1;
        d->cod_offset = (d->cod_offset << 1) | h264o_u(&d->br, 1);
    }
    d->info.n_bins++;
    return bin;
}
/* 9.3.3.2.3 DecodeBypass.  h264/cabac.go:468-481 (shifts by the bit instead of OR-ing it in:
 * Appendix A9). */
static int cabac_bypass(h264o_decoder *d) {
    d->cod_offset = (d->cod_offset << 1) | h264o_u(&d->br, 1);
    d->info.n_bins++;
    if (d->cod_offset >= d->cod_range) {
        d->cod_offset -= d->cod_range;
        return 1;
    }
    return 0;
}
/* 9.3.3.2.4 DecodeTerminate.  h264/cabac.go:486-499. */
static int cabac_terminate(h264o_decoder *d) {
    d->cod_range -= 2;
    d->info.n_bins++;
    if (d->cod_offset >= d->cod_range) return 1;
    while (d->cod_range < 256) {
        d->cod_range <<= 1;
        d->cod_offset = (d->cod_offset << 1) | h264o_u(&d->br, 1);
    }
    return 0;
}

int h264o_kat_cabac_bins(const uint8_t *bytes, size_t n, int pstate, int mps, int count, uint8_t *bins) {
    h264o_decoder *d = (h264o_decoder *)calloc(1, sizeof(*d));
    h264o_br_init(&d->br, bytes, n);
    h264o_cabac_init_engine(d);
    d->ctx[0] = (uint8_t)((pstate << 1) | mps);
    for (int i = 0; i < count; i++) bins[i] = (uint8_t)h264o_cabac_decision(d, 0);
    int st = d->ctx[0];
    free(d);
    return st;
}

/* ------------------------------------------------------------------ CABAC syntax elements 9.3.2 / 9.3.3.1 */
static int cabac_mb_skip_flag(h264o_decoder *d) { /* ctxIdxOffset 11 in P / SP slices, 24 in B slices (Table 9-34) */
    h264o_mb *a = MB_A(d), *b = MB_B(d);
    int inc = (a && !IS_SKIP(a->type)) + (b && !IS_SKIP(b->type));
    return h264o_cabac_decision(d, (d->sh.slice_type == 1 ? 24 : 11) + inc);
}

/* Table 9-36 I-slice mb_type bin string; `base` = 3 for I slices (bin 0 neighbour-dependent),
 * 17 for the intra suffix in P slices.  h264/cabac.go:181-277 (binIdxMbMap). */
static int cabac_intra_mb_type(h264o_decoder *d, int base, int islice) {
    if (islice) {
        h264o_mb *a = MB_A(d), *b = MB_B(d);
        int inc = (a && a->type != MBT_I4x4 && a->type != MBT_I8x8) + (b && b->type != MBT_I4x4 && b->type != MBT_I8x8);
        if (!h264o_cabac_decision(d, base + inc)) return 0;
        base += 2;
    } else if (!h264o_cabac_decision(d, base))
        return 0;
    if (cabac_terminate(d)) return 25;
    int t = 1;
    t += 12 * h264o_cabac_decision(d, base + 1);
    if (h264o_cabac_decision(d, base + 2)) t += 4 + 4 * h264o_cabac_decision(d, base + 2 + islice);
    t += 2 * h264o_cabac_decision(d, base + 3 + islice);
    t += h264o_cabac_decision(d, base + 3 + 2 * islice);
    return t;
}
/* Table 9-37 P mb_type; returns raw mb_type (0..4 inter, 5.. intra) */
static int cabac_p_mb_type(h264o_decoder *d) {
    if (!h264o_cabac_decision(d, 14)) {
        if (!h264o_cabac_decision(d, 15)) return 3 * h264o_cabac_decision(d, 16);
        return 2 - h264o_cabac_decision(d, 17);
    }
    return 5 + cabac_intra_mb_type(d, 17, 0);
}
/* Table 9-37 (b) B mb_type, ctxIdxOffset 27 (prefix) / 32 (intra suffix); returns raw mb_type 0..22, 23.. intra.
 * h264/cabac.go:278 leaves these bin strings as a TODO. */
static int cabac_b_mb_type(h264o_decoder *d) {
    h264o_mb *a = MB_A(d), *b = MB_B(d);
    int inc = (a && a->type != MBT_BSKIP && a->type != MBT_BDIRECT) + (b && b->type != MBT_BSKIP && b->type != MBT_BDIRECT);
    if (!h264o_cabac_decision(d, 27 + inc)) return 0; /* B_Direct_16x16 */
    if (!h264o_cabac_decision(d, 27 + 3)) return 1 + h264o_cabac_decision(d, 27 + 5); /* B_L0_16x16, B_L1_16x16 */
    int bits = h264o_cabac_decision(d, 27 + 4) << 3;
    bits |= h264o_cabac_decision(d, 27 + 5) << 2;
    bits |= h264o_cabac_decision(d, 27 + 5) << 1;
    bits |= h264o_cabac_decision(d, 27 + 5);
    if (bits < 8) return bits + 3; /* B_Bi_16x16 .. B_L1_L0_16x8 */
    if (bits == 13) return 23 + cabac_intra_mb_type(d, 32, 0);
    if (bits == 14) return 11; /* B_L1_L0_8x16 */
    if (bits == 15) return 22; /* B_8x8 */
    bits = (bits << 1) | h264o_cabac_decision(d, 27 + 5);
    return bits - 4; /* B_L0_Bi_16x8 .. B_Bi_Bi_8x16 */
}
/* Table 9-38 (b) B sub_mb_type, ctxIdxOffset 36 (h264/cabac.go:295 TODO) */
static int cabac_b_sub_mb_type(h264o_decoder *d) {
    if (!h264o_cabac_decision(d, 36)) return 0; /* B_Direct_8x8 */
    if (!h264o_cabac_decision(d, 37)) return 1 + h264o_cabac_decision(d, 39);
    int type = 3;
    if (h264o_cabac_decision(d, 38)) {
        if (h264o_cabac_decision(d, 39)) return 11 + h264o_cabac_decision(d, 39); /* B_L1_4x4, B_Bi_4x4 */
        type += 4;
    }
    type += 2 * h264o_cabac_decision(d, 39);
    type += h264o_cabac_decision(d, 39);
    return type;
}
static int cabac_p_sub_mb_type(h264o_decoder *d) {
    if (h264o_cabac_decision(d, 21)) return 0;
    if (!h264o_cabac_decision(d, 22)) return 1;
    if (h264o_cabac_decision(d, 23)) return 2;
    return 3;
}
static int cabac_transform8x8(h264o_decoder *d) {
    h264o_mb *a = MB_A(d), *b = MB_B(d);
    return h264o_cabac_decision(d, 399 + (a && a->t8x8) + (b && b->t8x8));
}
static int cabac_intra_chroma_mode(h264o_decoder *d) {
    h264o_mb *a = MB_A(d), *b = MB_B(d);
    int inc = (a && MB_IS_INTRA(a->type) && a->type != MBT_IPCM && a->chroma_mode != 0) +
              (b && MB_IS_INTRA(b->type) && b->type != MBT_IPCM && b->chroma_mode != 0);
    if (!h264o_cabac_decision(d, 64 + inc)) return 0;
    if (!h264o_cabac_decision(d, 64 + 3)) return 1;
    if (!h264o_cabac_decision(d, 64 + 3)) return 2;
    return 3;
}
static int cabac_cbp(h264o_decoder *d) {
    h264o_mb *a = MB_A(d), *b = MB_B(d);
    /* cbp of neighbours as seen by 9.3.3.1.1.4: unavailable / I_PCM behave as "all set";
     * P_Skip as 0 */
    int cbp_a = a ? (a->type == MBT_IPCM ? 0x2F : (a->cbp_luma | (a->cbp_chroma << 4))) : 0x0F;
    int cbp_b = b ? (b->type == MBT_IPCM ? 0x2F : (b->cbp_luma | (b->cbp_chroma << 4))) : 0x0F;
    int cbp = 0;
    for (int b8 = 0; b8 < 4; b8++) {
        int ca, cb; /* bit of the neighbouring 8x8 block: 1 -> condTermFlag 0 */
        if (b8 & 1)
            ca = (cbp >> (b8 - 1)) & 1;
        else
            ca = (cbp_a >> (b8 + 1)) & 1;
        if (b8 & 2)
            cb = (cbp >> (b8 - 2)) & 1;
        else
            cb = (cbp_b >> (b8 + 2)) & 1;
        int inc = (!ca) + 2 * (!cb);
        cbp |= h264o_cabac_decision(d, 73 + inc) << b8;
    }
    if (!d->asps->chroma_format_idc) return cbp; /* ChromaArrayType 0: the prefix only */
    /* chroma: condTermFlagN = N available && (I_PCM || cbp_chroma != 0) */
    int ca = a && (a->type == MBT_IPCM || a->cbp_chroma != 0), cb = b && (b->type == MBT_IPCM || b->cbp_chroma != 0);
    if (h264o_cabac_decision(d, 77 + ca + 2 * cb)) {
        ca = a && (a->type == MBT_IPCM || a->cbp_chroma == 2);
        cb = b && (b->type == MBT_IPCM || b->cbp_chroma == 2);
        cbp |= (1 + h264o_cabac_decision(d, 77 + 4 + ca + 2 * cb)) << 4;
    }
    return cbp;
}
static int cabac_mb_qp_delta(h264o_decoder *d) {
    int ctx = d->prev_dqp_nz ? 1 : 0, val = 0;
    while (h264o_cabac_decision(d, 60 + ctx)) {
        ctx = 2 + (ctx >> 1);
        val++;
        if (val > 104) return 0; /* bitstream error guard */
    }
    return (val & 1) ? (val + 1) >> 1 : -((val + 1) >> 1);
}
static int cabac_ref_idx(h264o_decoder *d, int list, int bx, int by) {
    /* 9.3.3.1.1.6: A/B partition refIdx > 0; partitions predicted in direct mode (B_Skip, B_Direct_16x16, B_Direct_8x8) count as 0 */
    int ia, ib;
    h264o_mb *a = luma_nb(d, bx - 1, by, &ia), *b = luma_nb(d, bx, by - 1, &ib);
    int a8 = (ia >> 3) * 2 + ((ia & 3) >> 1), b8 = (ib >> 3) * 2 + ((ib & 3) >> 1);
    int ra = (a && !((a->direct8 >> a8) & 1)) ? L_REF(a, list)[a8] : 0;
    int rb = (b && !((b->direct8 >> b8) & 1)) ? L_REF(b, list)[b8] : 0;
    int ctx = (ra > 0) + 2 * (rb > 0), ref = 0;
    while (h264o_cabac_decision(d, 54 + ctx)) {
        ref++;
        ctx = (ctx >> 2) + 4;
        if (ref > 32) return 0;
    }
    return ref;
}
/* 9.3.2.3 UEG3, signedValFlag=1, uCoff=9; ctxIdxInc per 9.3.3.1.1.7 */
static int cabac_mvd(h264o_decoder *d, int list, int comp, int bx, int by) {
    int ia, ib;
    h264o_mb *a = luma_nb(d, bx - 1, by, &ia), *b = luma_nb(d, bx, by - 1, &ib);
    int sum = (a ? L_MVD(a, list)[ia][comp] : 0) + (b ? L_MVD(b, list)[ib][comp] : 0);
    int base = comp ? 47 : 40;
    if (!h264o_cabac_decision(d, base + (sum > 2) + (sum > 32))) return 0;
    int v = 1, ctx = base + 3;
    while (v < 9 && h264o_cabac_decision(d, ctx)) {
        if (v < 4) ctx++;
        v++;
    }
    if (v >= 9) {
        int k = 3;
        while (cabac_bypass(d)) {
            v += 1 << k;
            k++;
            if (k > 24) return 0;
        }
        while (k--) v += cabac_bypass(d) << k;
    }
    return cabac_bypass(d) ? -v : v;
}

static const int sig_off[5] = {0, 15, 29, 44, 47};
static const int abs_off[5] = {0, 10, 20, 30, 39};

/* 7.3.5.3.3 residual_block_cabac(); cbf_inc < 0: coded_block_flag not parsed (inferred 1). */
static int cabac_residual(h264o_decoder *d, int16_t *coef, int cat, int maxnum, int cbf_inc) {
    memset(coef, 0, sizeof(int16_t) * maxnum);
    if (cbf_inc >= 0 && !h264o_cabac_decision(d, 85 + cat * 4 + cbf_inc)) return 0;
    int pos[64], n = 0, last = 0;
    for (int i = 0; i < maxnum - 1; i++) {
        int sctx, lctx;
        /* field-coded blocks (field pictures, h264/slice.go:867-872) have significance contexts of their own: ctxIdxOffset 277 / 338 (436 / 451 for 8x8 blocks) and,
         * for 8x8 blocks, the field column of Table 9-43 */
        if (cat == 5) {
            sctx = (d->field_pic ? 436 + h264o_sig8x8_field_ctx[i] : 402 + h264o_sig8x8_ctx[i]);
            lctx = (d->field_pic ? 451 : 417) + h264o_last8x8_ctx[i];
        } else {
            int inc = cat == 3 ? (i < 2 ? i : 2) : i;
            sctx = (d->field_pic ? 277 : 105) + sig_off[cat] + inc;
            lctx = (d->field_pic ? 338 : 166) + sig_off[cat] + inc;
        }
        if (h264o_cabac_decision(d, sctx)) {
            pos[n++] = i;
            if (h264o_cabac_decision(d, lctx)) {
                last = 1;
                break;
            }
        }
    }
    if (!last) pos[n++] = maxnum - 1;
    int eq1 = 0, gt1 = 0;
    int base = cat == 5 ? 426 : 227 + abs_off[cat];
    for (int k = n - 1; k >= 0; k--) {
        int inc0 = gt1 ? 0 : (1 + eq1 < 4 ? 1 + eq1 : 4);
        int a;
        if (!h264o_cabac_decision(d, base + inc0)) {
            a = 1;
            eq1++;
        } else {
            int lim = 4 - (cat == 3);
            int inc = 5 + (gt1 < lim ? gt1 : lim);
            a = 2;
            while (a < 15 && h264o_cabac_decision(d, base + inc)) a++;
            if (a >= 15) {
                int kk = 0;
                while (cabac_bypass(d)) {
                    a += 1 << kk;
                    kk++;
                    if (kk > 24) return h264o_fail(d, "cabac: abs level escape overflow");
                }
                while (kk--) a += cabac_bypass(d) << kk;
            }
            gt1++;
        }
        coef[pos[k]] = (int16_t)(cabac_bypass(d) ? -a : a);
    }
    return n;
}

/* coded_block_flag ctxIdxInc (9.3.3.1.1.9) */
static int cbf_inc_from(h264o_decoder *d, h264o_mb *a, int fa, h264o_mb *b, int fb) {
    int cur_intra = MB_IS_INTRA(d->c.type);
    int ca = a ? fa : cur_intra, cb = b ? fb : cur_intra;
    return ca + 2 * cb;
}
static int cbf_inc_luma(h264o_decoder *d, int bx, int by) {
    int ia, ib;
    h264o_mb *a = luma_nb(d, bx - 1, by, &ia), *b = luma_nb(d, bx, by - 1, &ib);
    return cbf_inc_from(d, a, a ? a->nnz[ia] != 0 : 0, b, b ? b->nnz[ib] != 0 : 0);
}
static int cbf_inc_chroma_ac(h264o_decoder *d, int c, int cx, int cy) {
    int ia, ib;
    h264o_mb *a = chroma_nb(d, cx - 1, cy, &ia), *b = chroma_nb(d, cx, cy - 1, &ib);
    return cbf_inc_from(d, a, a ? a->nnz[16 + 4 * c + ia] != 0 : 0, b, b ? b->nnz[16 + 4 * c + ib] != 0 : 0);
}
static int cbf_inc_dc(h264o_decoder *d, int bit) {
    h264o_mb *a = MB_A(d), *b = MB_B(d);
    return cbf_inc_from(d, a, a ? (a->cbf_dc >> bit) & 1 : 0, b, b ? (b->cbf_dc >> bit) & 1 : 0);
}

/* ------------------------------------------------------------------ residual() 7.3.5.3 */
static int parse_residual(h264o_decoder *d) {
    h264o_curmb *c = &d->c;
    h264o_mb *m = CUR(d);
    int cabac = d->apps->entropy_coding_mode_flag;
    int i16 = c->type == MBT_I16x16;
    if (i16) {
        int n;
        if (cabac)
            n = cabac_residual(d, c->i16dc, 0, 16, cbf_inc_dc(d, 0));
        else
            n = cavlc_residual(d, c->i16dc, 16, cavlc_nc_luma(d, 0, 0));
        if (n < 0) return n;
        if (n > 0) m->cbf_dc |= 1;
    }
    for (int b8 = 0; b8 < 4; b8++) {
        if (!(c->cbp_luma & (1 << b8))) continue;
        if (c->t8x8 && cabac) {
            int n = cabac_residual(d, c->luma8[b8], 5, 64, -1);
            if (n < 0) return n;
            int bx = (b8 & 1) * 2, by = (b8 >> 1) * 2;
            m->nnz[by * 4 + bx] = m->nnz[by * 4 + bx + 1] = m->nnz[by * 4 + 4 + bx] = m->nnz[by * 4 + 5 + bx] = (uint8_t)n;
            if (n) m->nzmask |= (uint16_t)(0x33 << (by * 4 + bx));
            continue;
        }
        int any = 0;
        for (int b4 = 0; b4 < 4; b4++) {
            int idx = b8 * 4 + b4, r = h264o_blk_raster(idx), bx = r & 3, by = r >> 2, n;
            if (i16) {
                c->luma[idx][0] = 0;
                if (cabac)
                    n = cabac_residual(d, c->luma[idx] + 1, 1, 15, cbf_inc_luma(d, bx, by));
                else
                    n = cavlc_residual(d, c->luma[idx] + 1, 15, cavlc_nc_luma(d, bx, by));
            } else if (cabac)
                n = cabac_residual(d, c->luma[idx], 2, 16, cbf_inc_luma(d, bx, by));
            else
                n = cavlc_residual(d, c->luma[idx], 16, cavlc_nc_luma(d, bx, by));
            if (n < 0) return n;
            m->nnz[r] = (uint8_t)n;
            if (n) {
                m->nzmask |= (uint16_t)(1 << r);
                any = 1;
            }
            if (c->t8x8) /* CAVLC 8x8: the four 4x4 reads interleave into one 8x8 block (7.3.5.3.2) */
                for (int i = 0; i < 16; i++) c->luma8[b8][4 * i + b4] = c->luma[idx][i];
        }
        if (c->t8x8 && any) {
            int bx = (b8 & 1) * 2, by = (b8 >> 1) * 2;
            m->nzmask |= (uint16_t)(0x33 << (by * 4 + bx));
        }
    }
    if (c->cbp_chroma) {
        for (int cc = 0; cc < 2; cc++) {
            int n;
            if (cabac)
                n = cabac_residual(d, c->cdc[cc], 3, 4, cbf_inc_dc(d, 1 + cc));
            else
                n = cavlc_residual(d, c->cdc[cc], 4, -1);
            if (n < 0) return n;
            if (n > 0) m->cbf_dc |= (uint8_t)(2 << cc);
        }
    }
    if (c->cbp_chroma & 2) {
        for (int cc = 0; cc < 2; cc++)
            for (int b4 = 0; b4 < 4; b4++) {
                int cx = b4 & 1, cy = b4 >> 1, n;
                c->cac[cc][b4][0] = 0;
                if (cabac)
                    n = cabac_residual(d, c->cac[cc][b4] + 1, 4, 15, cbf_inc_chroma_ac(d, cc, cx, cy));
                else
                    n = cavlc_residual(d, c->cac[cc][b4] + 1, 15, cavlc_nc_chroma(d, cc, cx, cy));
                if (n < 0) return n;
                m->nnz[16 + 4 * cc + b4] = (uint8_t)n;
            }
    }
    return 0;
}

/* ------------------------------------------------------------------ intra pred mode derivation 8.3.1.1 / 8.3.2.1 */
static int pred_intra_mode(h264o_decoder *d, int bx, int by) {
    int ia, ib;
    h264o_mb *a = luma_nb(d, bx - 1, by, &ia), *b = luma_nb(d, bx, by - 1, &ib);
    int cip = d->apps->constrained_intra_pred_flag;
    if (!a || !b || (cip && (MB_IS_INTER(a->type) || MB_IS_INTER(b->type)))) return 2; /* dcPredModePredictedFlag */
    int ma = (a->type == MBT_I4x4 || a->type == MBT_I8x8) ? a->ipm[ia] : 2;
    int mb = (b->type == MBT_I4x4 || b->type == MBT_I8x8) ? b->ipm[ib] : 2;
    return ma < mb ? ma : mb;
}

/* ------------------------------------------------------------------ motion vector prediction 8.4.1.3 */
typedef struct {
    int avail; /* partition available (8.4.1.3.2) */
    int ref;   /* -1: intra / not available / list not used */
    int mv[2];
} nbmv;

/* neighbouring partition covering block (bx, by) (relative to the current MB, -1 / 4 = outside) as seen by list `list`.
 * Inside the current macroblock a block is available once its motion for that list is final (cur_done[list]) and -- direct
 * sub-macroblocks are derived up front -- only if it lies in a sub-macroblock not later than the one being decoded (6.4.11.7). */
static void get_nbmv(h264o_decoder *d, int list, int bx, int by, nbmv *o) {
    h264o_curmb *c = &d->c;
    h264o_mb *m = NULL;
    o->avail = 0;
    o->ref = -1;
    o->mv[0] = o->mv[1] = 0;
    if (by >= 0 && bx >= 4) return; /* right MB: never available */
    if (bx >= 0 && bx < 4 && by >= 0) {
        if (!((d->cur_done[list] >> (by * 4 + bx)) & 1)) return;
        if ((by >> 1) * 2 + (bx >> 1) > d->cur_sub) return;
        m = CUR(d);
    } else {
        int mx = c->mbx + (bx < 0 ? -1 : (bx >= 4 ? 1 : 0)), my = c->mby + (by < 0 ? -1 : 0);
        m = mb_at(d, mx, my);
        if (!m) return;
        bx &= 3;
        by &= 3;
    }
    o->avail = 1;
    if (MB_IS_INTRA(m->type)) return;
    o->ref = L_REF(m, list)[(by >> 1) * 2 + (bx >> 1)];
    if (o->ref < 0) return; /* the partition does not use this list: refIdx -1, mv 0 */
    o->mv[0] = L_MV(m, list)[by * 4 + bx][0];
    o->mv[1] = L_MV(m, list)[by * 4 + bx][1];
}
static int median3(int a, int b, int c) {
    int mn = a < b ? a : b, mx = a < b ? b : a;
    return c < mn ? mn : (c > mx ? mx : c);
}
/* shape: 0 = median only, 1 = 16x8 top, 2 = 16x8 bottom, 3 = 8x16 left, 4 = 8x16 right */
static void predict_mv(h264o_decoder *d, int list, int bx, int by, int w, int ref, int shape, int mvp[2]) {
    nbmv A, B, C;
    get_nbmv(d, list, bx - 1, by, &A);
    get_nbmv(d, list, bx, by - 1, &B);
    get_nbmv(d, list, bx + w, by - 1, &C);
    if (!C.avail) get_nbmv(d, list, bx - 1, by - 1, &C);
    if ((shape == 1 && B.ref == ref) || (shape == 4 && C.ref == ref)) {
        const nbmv *s = shape == 1 ? &B : &C;
        mvp[0] = s->mv[0];
        mvp[1] = s->mv[1];
        return;
    }
    if ((shape == 2 || shape == 3) && A.ref == ref) {
        mvp[0] = A.mv[0];
        mvp[1] = A.mv[1];
        return;
    }
    if (!B.avail && !C.avail && A.avail) {
        B = A;
        C = A;
    }
    int na = A.ref == ref, nb = B.ref == ref, nc = C.ref == ref;
    if (na + nb + nc == 1) {
        const nbmv *s = na ? &A : (nb ? &B : &C);
        mvp[0] = s->mv[0];
        mvp[1] = s->mv[1];
        return;
    }
    mvp[0] = median3(A.mv[0], B.mv[0], C.mv[0]);
    mvp[1] = median3(A.mv[1], B.mv[1], C.mv[1]);
}
static void set_part(h264o_decoder *d, int list, int bx, int by, int w, int h, const int mv[2], const int mvd[2]) {
    h264o_mb *m = CUR(d);
    for (int y = by; y < by + h; y++)
        for (int x = bx; x < bx + w; x++) {
            L_MV(m, list)[y * 4 + x][0] = (int16_t)mv[0];
            L_MV(m, list)[y * 4 + x][1] = (int16_t)mv[1];
            L_MVD(m, list)[y * 4 + x][0] = (int16_t)abs(mvd[0]);
            L_MVD(m, list)[y * 4 + x][1] = (int16_t)abs(mvd[1]);
            d->cur_done[list] |= (uint16_t)(1 << (y * 4 + x));
        }
}
static void read_mvd(h264o_decoder *d, int list, int bx, int by, int mvd[2]) {
    if (d->apps->entropy_coding_mode_flag) {
        mvd[0] = cabac_mvd(d, list, 0, bx, by);
        mvd[1] = cabac_mvd(d, list, 1, bx, by);
    } else {
        mvd[0] = h264o_se(&d->br);
        mvd[1] = h264o_se(&d->br);
    }
}
static void do_part(h264o_decoder *d, int list, int bx, int by, int w, int h, int shape) {
    h264o_mb *m = CUR(d);
    int mvd[2], mvp[2], mv[2];
    read_mvd(d, list, bx, by, mvd);
    predict_mv(d, list, bx, by, w, L_REF(m, list)[(by >> 1) * 2 + (bx >> 1)], shape, mvp);
    mv[0] = mvp[0] + mvd[0];
    mv[1] = mvp[1] + mvd[1];
    set_part(d, list, bx, by, w, h, mv, mvd);
}
static int read_ref_idx(h264o_decoder *d, int list, int bx, int by) {
    int nref = list ? d->sh.num_ref_idx_l1_active_minus1 : d->sh.num_ref_idx_l0_active_minus1;
    if (nref == 0) return 0;
    const int r = d->apps->entropy_coding_mode_flag ? cabac_ref_idx(d, list, bx, by) : (int)h264o_te(&d->br, nref);
    if (r < 0 || r > nref) { /* 7.4.5.1: 0 .. num_ref_idx_active_minus1 (damaged streams) */
        h264o_fail(d, "ref_idx_l%d %d beyond the %d active entries", list, r, nref + 1);
        return 0;
    }
    return r;
}
static void set_refids(h264o_decoder *d, h264o_mb *m) {
    for (int l = 0; l < 2; l++)
        for (int i = 0; i < 4; i++) {
            int r = L_REF(m, l)[i];
            h264o_pic *p = (r >= 0 && r <= 32) ? (l ? d->rpl1[r] : d->rpl0[r]) : NULL;
            L_REFID(m, l)[i] = p ? p->id : -1;
        }
}

/* 8.4.1.1 P_Skip motion */
static void pskip_motion(h264o_decoder *d) {
    h264o_mb *m = CUR(d);
    nbmv A, B;
    int mv[2] = {0, 0}, zero[2] = {0, 0};
    memset(m->ref, 0, sizeof(m->ref));
    get_nbmv(d, 0, -1, 0, &A);
    get_nbmv(d, 0, 0, -1, &B);
    if (A.avail && B.avail && !(A.ref == 0 && A.mv[0] == 0 && A.mv[1] == 0) && !(B.ref == 0 && B.mv[0] == 0 && B.mv[1] == 0))
        predict_mv(d, 0, 0, 0, 4, 0, 0, mv);
    set_part(d, 0, 0, 0, 4, 4, mv, zero);
    set_refids(d, m);
}

/* ------------------------------------------------------------------ direct prediction 8.4.1.2 (B_Skip, B_Direct_16x16, B_Direct_8x8) */
/* co-located 4x4 block of block (bx, by): 8.4.1.2.1 for frame pictures.  With direct_8x8_inference_flag the corner block
 * of the 8x8 quadrant stands for the whole quadrant.  Returns refIdxCol (-1: intra), the vector and the picture it points to. */
static int colocated(h264o_decoder *d, int bx, int by, int mvcol[2], int *refid_col) {
    const h264o_pic *col = d->rpl1[0];
    mvcol[0] = mvcol[1] = 0;
    *refid_col = -1;
    if (!col || !col->mbs || col->n_mbs != d->wmb * d->hmb) return -1;
    const h264o_mb *cm = &col->mbs[d->c.addr];
    if (d->asps->direct_8x8_inference_flag) bx = (bx >> 1) * 3, by = (by >> 1) * 3;
    if (!MB_IS_INTER(cm->type)) return -1;
    int i8 = (by >> 1) * 2 + (bx >> 1), l = cm->ref[i8] >= 0 ? 0 : 1;
    if (L_REF(cm, l)[i8] < 0) return -1;
    mvcol[0] = L_MV(cm, l)[by * 4 + bx][0];
    mvcol[1] = L_MV(cm, l)[by * 4 + bx][1];
    *refid_col = L_REFID(cm, l)[i8];
    return L_REF(cm, l)[i8];
}
static int min_positive(int a, int b) { return (a >= 0 && b >= 0) ? (a < b ? a : b) : (a > b ? a : b); }

/* derive the motion of the 8x8 quadrants in `mask8` (bit i = quadrant i) */
static int direct_pred(h264o_decoder *d, int mask8) {
    h264o_mb *m = CUR(d);
    const int zero[2] = {0, 0};
    if (d->sh.direct_spatial_mv_pred_flag) { /* 8.4.1.2.2 */
        int ref[2], mvp[2][2] = {{0, 0}, {0, 0}};
        int save_sub = d->cur_sub;
        d->cur_sub = -1; /* the neighbours A, B, C of the MACROBLOCK: nothing inside it counts */
        for (int l = 0; l < 2; l++) {
            nbmv A, B, C;
            get_nbmv(d, l, -1, 0, &A);
            get_nbmv(d, l, 0, -1, &B);
            get_nbmv(d, l, 4, -1, &C);
            if (!C.avail) get_nbmv(d, l, -1, -1, &C);
            ref[l] = min_positive(A.ref, min_positive(B.ref, C.ref));
        }
        if (ref[0] < 0 && ref[1] < 0)
            ref[0] = ref[1] = 0; /* directZeroPredictionFlag: both vectors stay zero */
        else
            for (int l = 0; l < 2; l++)
                if (ref[l] >= 0) predict_mv(d, l, 0, 0, 4, ref[l], 0, mvp[l]);
        d->cur_sub = save_sub;
        const int col_short = d->rpl1[0] && d->rpl1[0]->ref == 1;
        for (int i8 = 0; i8 < 4; i8++) {
            if (!((mask8 >> i8) & 1)) continue;
            for (int k = 0; k < 4; k++) {
                int bx = (i8 & 1) * 2 + (k & 1), by = (i8 >> 1) * 2 + (k >> 1), mvcol[2], rid;
                int refcol = colocated(d, bx, by, mvcol, &rid);
                int colzero = col_short && refcol == 0 && mvcol[0] >= -1 && mvcol[0] <= 1 && mvcol[1] >= -1 && mvcol[1] <= 1;
                for (int l = 0; l < 2; l++) {
                    L_REF(m, l)[i8] = (int8_t)ref[l];
                    const int *mv = (ref[l] < 0 || (ref[l] == 0 && colzero)) ? zero : mvp[l];
                    set_part(d, l, bx, by, 1, 1, mv, zero);
                }
            }
        }
        return 0;
    }
    /* 8.4.1.2.3 temporal direct */
    for (int i8 = 0; i8 < 4; i8++) {
        if (!((mask8 >> i8) & 1)) continue;
        for (int k = 0; k < 4; k++) {
            int bx = (i8 & 1) * 2 + (k & 1), by = (i8 >> 1) * 2 + (k >> 1), mvcol[2], rid;
            int refcol = colocated(d, bx, by, mvcol, &rid);
            int ref0 = 0;
            if (refcol >= 0) { /* the picture the co-located block refers to, as an index of the current RefPicList0 */
                ref0 = -1;
                for (int i = 0; i <= d->sh.num_ref_idx_l0_active_minus1 && ref0 < 0; i++)
                    if (d->rpl0[i] && d->rpl0[i]->id == rid) ref0 = i;
                if (ref0 < 0) return h264o_fail(d, "temporal direct: the co-located reference is not in RefPicList0");
            }
            const h264o_pic *p0 = d->rpl0[ref0], *p1 = d->rpl1[0];
            if (!p0 || !p1) return h264o_fail(d, "temporal direct without reference pictures");
            int mv0[2], mv1[2];
            int tb = h264o_clip3(-128, 127, d->cur->poc - p0->poc), td = h264o_clip3(-128, 127, p1->poc - p0->poc);
            if (p0->ref == 2 || td == 0) {
                mv0[0] = mvcol[0], mv0[1] = mvcol[1];
                mv1[0] = mv1[1] = 0;
            } else {
                int tx = (16384 + abs(td / 2)) / td;
                int dsf = h264o_clip3(-1024, 1023, (tb * tx + 32) >> 6);
                for (int c = 0; c < 2; c++) {
                    mv0[c] = (dsf * mvcol[c] + 128) >> 8;
                    mv1[c] = mv0[c] - mvcol[c];
                }
            }
            m->ref[i8] = (int8_t)ref0;
            m->ref1[i8] = 0;
            set_part(d, 0, bx, by, 1, 1, mv0, zero);
            set_part(d, 1, bx, by, 1, 1, mv1, zero);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ macroblock_layer() 7.3.5 */
static void trace_mb(h264o_decoder *d, h264o_mb *m) {
    if (!d->trace || d->trace_pos >= d->trace_cap) return;
    int32_t *t = d->trace + 8 * d->trace_pos++;
    t[0] = d->c.mb_type_raw;
    t[1] = m->cbp_luma | (m->cbp_chroma << 4);
    t[2] = m->qp;
    t[3] = d->c.type == MBT_I16x16 ? d->c.i16mode : m->chroma_mode;
    t[4] = m->t8x8;
    t[5] = m->mv[0][0];
    t[6] = m->mv[0][1];
    t[7] = m->ref[0];
    if (d->sh.slice_type == 1) t[7] -= 100; /* marks macroblocks of B slices (raw mb_type is per slice type) */
}

static void begin_mb(h264o_decoder *d, int addr) {
    h264o_curmb *c = &d->c;
    h264o_mb *m = &d->mb[addr];
    memset(c, 0, sizeof(*c));
    c->addr = addr;
    c->mbx = addr % d->wmb;
    c->mby = addr / d->wmb;
    memset(m, 0, sizeof(*m));
    m->slice_id = (uint16_t)d->slice_id;
    memset(m->ipm, -1, sizeof(m->ipm));
    memset(m->ref, -1, sizeof(m->ref));
    memset(m->ref1, -1, sizeof(m->ref1));
    for (int i = 0; i < 4; i++) m->refid[i] = m->refid1[i] = -1;
    m->alpha_off = (int8_t)(d->sh.slice_alpha_c0_offset_div2 * 2);
    m->beta_off = (int8_t)(d->sh.slice_beta_offset_div2 * 2);
    m->dbf_idc = (uint8_t)d->sh.disable_deblocking_filter_idc;
    d->cur_done[0] = d->cur_done[1] = 0;
    d->cur_sub = 3;
}
static void finish_mb(h264o_decoder *d) {
    h264o_mb *m = CUR(d);
    m->type = (uint8_t)d->c.type;
    m->qp = (uint8_t)d->qp;
    for (int cc = 0; cc < 2; cc++) {
        int off = cc ? d->apps->second_chroma_qp_index_offset : d->apps->chroma_qp_index_offset;
        m->qpc[cc] = (uint8_t)h264o_qpc(h264o_clip3(0, 51, d->qp + off));
    }
    if (m->type == MBT_IPCM) m->qp = 0, m->qpc[0] = m->qpc[1] = (uint8_t)h264o_qpc(h264o_clip3(0, 51, d->apps->chroma_qp_index_offset));
    if (m->type == MBT_IPCM) m->qpc[1] = (uint8_t)h264o_qpc(h264o_clip3(0, 51, d->apps->second_chroma_qp_index_offset));
    h264o_recon_mb(d, m);
    trace_mb(d, m);
    d->info.n_mbs++;
}

static int decode_pskip(h264o_decoder *d, int addr) {
    begin_mb(d, addr);
    d->c.type = MBT_PSKIP;
    d->c.mb_type_raw = -1;
    CUR(d)->type = MBT_PSKIP; /* so that set_part/refs see an inter MB */
    pskip_motion(d);
    d->prev_dqp_nz = 0;
    finish_mb(d);
    return 0;
}

/* B_Skip: direct prediction of the whole macroblock, no residual (7.3.4, 8.4.1.2) */
static int decode_bskip(h264o_decoder *d, int addr) {
    begin_mb(d, addr);
    d->c.type = MBT_BSKIP;
    d->c.mb_type_raw = -1;
    CUR(d)->type = MBT_BSKIP;
    CUR(d)->direct8 = 15;
    if (direct_pred(d, 15) < 0) return -1;
    set_refids(d, CUR(d));
    d->prev_dqp_nz = 0;
    finish_mb(d);
    return 0;
}

/* Table 7-14: prediction modes of the two partitions of mb_type 4..21 (1 = Pred_L0, 2 = Pred_L1, 3 = BiPred) */
static const uint8_t b_part_modes[9][2] = {{1, 1}, {2, 2}, {1, 2}, {2, 1}, {1, 3}, {2, 3}, {3, 1}, {3, 2}, {3, 3}};
/* Table 7-18: sub_mb_type of B macroblocks -> prediction mode (0 = direct) and sub-partition shape (0 8x8, 1 8x4, 2 4x8, 3 4x4) */
static const uint8_t b_sub_mode[13] = {0, 1, 2, 3, 1, 1, 2, 2, 3, 3, 1, 2, 3};
static const uint8_t b_sub_shape[13] = {0, 0, 0, 0, 1, 2, 1, 2, 1, 2, 3, 3, 3};

typedef struct {
    int bx, by, w, h, shape, mode;
} ipart;

/* A partition that does not use list `list`: for the partitions decoded after it, it is an AVAILABLE neighbour with
 * refIdxLX = -1 and a zero vector (8.4.1.3.2); availability is a matter of decoding order only (6.4.11.7). */
static void mark_unused(h264o_decoder *d, int list, int bx, int by, int w, int h) {
    for (int y = by; y < by + h; y++)
        for (int x = bx; x < bx + w; x++) d->cur_done[list] |= (uint16_t)(1 << (y * 4 + x));
}
/* mb_pred() for the unsplit / two-partition inter types of P and B slices: every ref_idx_l0, every ref_idx_l1, every
 * mvd_l0, every mvd_l1 (7.3.5.1) */
static void decode_parts(h264o_decoder *d, const ipart *pt, int n) {
    h264o_mb *m = CUR(d);
    for (int l = 0; l < 2; l++)
        for (int i = 0; i < n; i++) {
            if (!((pt[i].mode >> l) & 1)) continue;
            int r = read_ref_idx(d, l, pt[i].bx, pt[i].by);
            for (int y = pt[i].by; y < pt[i].by + pt[i].h; y += 2)
                for (int x = pt[i].bx; x < pt[i].bx + pt[i].w; x += 2) L_REF(m, l)[(y >> 1) * 2 + (x >> 1)] = (int8_t)r;
        }
    for (int l = 0; l < 2; l++)
        for (int i = 0; i < n; i++) {
            if ((pt[i].mode >> l) & 1)
                do_part(d, l, pt[i].bx, pt[i].by, pt[i].w, pt[i].h, pt[i].shape);
            else
                mark_unused(d, l, pt[i].bx, pt[i].by, pt[i].w, pt[i].h);
        }
}

static int decode_mb(h264o_decoder *d, int addr) {
    h264o_curmb *c;
    h264o_mb *m;
    h264o_br *b = &d->br;
    int cabac = d->apps->entropy_coding_mode_flag;
    int islice = d->sh.slice_type == 2;
    begin_mb(d, addr);
    c = &d->c;
    m = CUR(d);
    const int bslice = d->sh.slice_type == 1;
    int raw = cabac ? (islice ? cabac_intra_mb_type(d, 3, 1) : (bslice ? cabac_b_mb_type(d) : cabac_p_mb_type(d))) : (int)h264o_ue(b);
    c->mb_type_raw = raw;
    int it = islice ? raw : raw - (bslice ? 23 : 5); /* intra mb_type (Table 7-11) when >= 0 */
    if (bslice && raw < 23) { /* Table 7-14 */
        c->type = raw == 0 ? MBT_BDIRECT : (raw <= 3 ? MBT_P16x16 : (raw == 22 ? MBT_P8x8 : ((raw & 1) ? MBT_P8x16 : MBT_P16x8)));
    } else if (!islice && !bslice && raw < 5) {
        static const int pt[5] = {MBT_P16x16, MBT_P16x8, MBT_P8x16, MBT_P8x8, MBT_P8x8};
        c->type = pt[raw];
    } else if (it == 0)
        c->type = MBT_I4x4;
    else if (it >= 1 && it <= 24) {
        c->type = MBT_I16x16;
        c->i16mode = (it - 1) & 3;
        c->cbp_chroma = ((it - 1) >> 2) % 3;
        c->cbp_luma = it >= 13 ? 15 : 0;
        if (c->cbp_chroma && !d->asps->chroma_format_idc) return h264o_fail(d, "mb %d: Intra16x16 mb_type with chroma coefficients in a monochrome stream", addr);
    } else if (it == 25)
        c->type = MBT_IPCM;
    else
        return h264o_fail(d, "mb %d: bad mb_type %d", addr, raw);
    m->type = (uint8_t)c->type; /* provisional: neighbour helpers look at CUR() */

    if (c->type == MBT_IPCM) {
        /* After the terminate bin (binVal 1) the engine has consumed every bit the encoder's
         * flush wrote, including its final '1' (9.3.4.5): 9 + #renorm-shifts bits were read for
         * (#shifts - 1) + 7 + 1 + 2 written.  pcm_alignment_zero_bit(s) follow directly. */
        while (b->pos & 7) h264o_u(b, 1);
        const int mono = !d->asps->chroma_format_idc; /* 256 luma samples only; the chroma samples of the reconstruction are 128 */
        for (int i = 0; i < 384; i++) c->pcm[i] = mono && i >= 256 ? 128 : (uint8_t)h264o_u(b, 8);
        if (cabac) h264o_cabac_init_engine(d);
        memset(m->nnz, 16, sizeof(m->nnz));
        m->nzmask = 0xFFFF;
        m->cbf_dc = 7;
        m->cbp_luma = 15;
        m->cbp_chroma = 2;
        d->prev_dqp_nz = 0;
        finish_mb(d); /* QP_Y carries over unchanged (mb_qp_delta inferred 0); deblocking uses qP 0 */
        return b->err ? h264o_fail(d, "mb %d: pcm overrun", addr) : 0;
    }

    int no_sub8 = 1; /* NoSubMbPartSizeLessThan8x8Flag (7.3.5) */
    if (c->type == MBT_BDIRECT) {
        m->direct8 = 15;
        if (direct_pred(d, 15) < 0) return -1;
        if (!d->asps->direct_8x8_inference_flag) no_sub8 = 0;
    } else if (c->type == MBT_P8x8) {
        int mode[4], shape[4];
        for (int i = 0; i < 4; i++) {
            int st = cabac ? (bslice ? cabac_b_sub_mb_type(d) : cabac_p_sub_mb_type(d)) : (int)h264o_ue(b);
            if (st > (bslice ? 12 : 3)) return h264o_fail(d, "mb %d: bad sub_mb_type", addr);
            c->sub_type[i] = st;
            mode[i] = bslice ? b_sub_mode[st] : 1;
            shape[i] = bslice ? b_sub_shape[st] : st;
            if (mode[i] == 0) {
                m->direct8 |= (uint8_t)(1 << i);
                if (!d->asps->direct_8x8_inference_flag) no_sub8 = 0;
            } else if (shape[i] != 0)
                no_sub8 = 0;
        }
        if (m->direct8) { /* direct sub-macroblocks are derived first: later sub-macroblocks predict from them */
            d->cur_sub = -1;
            if (direct_pred(d, m->direct8) < 0) return -1;
        }
        for (int l = 0; l < 2; l++)
            for (int i = 0; i < 4; i++) {
                if (!((mode[i] >> l) & 1)) continue;
                d->cur_sub = i;
                L_REF(m, l)[i] = (!bslice && raw == 4) ? 0 : (int8_t)read_ref_idx(d, l, (i & 1) * 2, (i >> 1) * 2);
            }
        for (int l = 0; l < 2; l++)
            for (int i = 0; i < 4; i++) {
                int bx = (i & 1) * 2, by = (i >> 1) * 2;
                if (!((mode[i] >> l) & 1)) {
                    if (mode[i]) mark_unused(d, l, bx, by, 2, 2); /* (direct sub-macroblocks are final already) */
                    continue;
                }
                d->cur_sub = i;
                switch (shape[i]) {
                case 0: do_part(d, l, bx, by, 2, 2, 0); break;
                case 1: do_part(d, l, bx, by, 2, 1, 0); do_part(d, l, bx, by + 1, 2, 1, 0); break;
                case 2: do_part(d, l, bx, by, 1, 2, 0); do_part(d, l, bx + 1, by, 1, 2, 0); break;
                default:
                    do_part(d, l, bx, by, 1, 1, 0);
                    do_part(d, l, bx + 1, by, 1, 1, 0);
                    do_part(d, l, bx, by + 1, 1, 1, 0);
                    do_part(d, l, bx + 1, by + 1, 1, 1, 0);
                }
            }
        d->cur_sub = 3;
    } else if (MB_IS_INTER(c->type)) {
        ipart pt[2];
        int n, m0 = 1, m1 = 1;
        if (bslice) {
            if (raw <= 3)
                m0 = raw;
            else
                m0 = b_part_modes[(raw - 4) >> 1][0], m1 = b_part_modes[(raw - 4) >> 1][1];
        }
        if (c->type == MBT_P16x16)
            pt[0] = (ipart){0, 0, 4, 4, 0, m0}, n = 1;
        else if (c->type == MBT_P16x8)
            pt[0] = (ipart){0, 0, 4, 2, 1, m0}, pt[1] = (ipart){0, 2, 4, 2, 2, m1}, n = 2;
        else
            pt[0] = (ipart){0, 0, 2, 4, 3, m0}, pt[1] = (ipart){2, 0, 2, 4, 4, m1}, n = 2;
        decode_parts(d, pt, n);
    } else {
        /* intra: transform_size_8x8_flag, pred modes, chroma mode (7.3.5, 7.3.5.1) */
        if (c->type == MBT_I4x4 && d->apps->transform_8x8_mode_flag) {
            c->t8x8 = cabac ? cabac_transform8x8(d) : (int)h264o_u(b, 1);
            if (c->t8x8) c->type = MBT_I8x8, m->type = MBT_I8x8;
            m->t8x8 = (uint8_t)c->t8x8;
        }
        if (c->type == MBT_I4x4 || c->type == MBT_I8x8) {
            int n = c->type == MBT_I8x8 ? 4 : 16;
            for (int i = 0; i < n; i++) {
                int r = n == 4 ? ((i >> 1) * 8 + (i & 1) * 2) : h264o_blk_raster(i);
                int bx = r & 3, by = r >> 2;
                int pred = pred_intra_mode(d, bx, by), mode;
                int flag = cabac ? h264o_cabac_decision(d, 68) : (int)h264o_u(b, 1);
                if (flag)
                    mode = pred;
                else {
                    int rem;
                    if (cabac) {
                        rem = h264o_cabac_decision(d, 69);
                        rem |= h264o_cabac_decision(d, 69) << 1;
                        rem |= h264o_cabac_decision(d, 69) << 2;
                    } else
                        rem = (int)h264o_u(b, 3);
                    mode = rem < pred ? rem : rem + 1;
                }
                m->ipm[r] = (int8_t)mode;
                if (n == 4) m->ipm[r + 1] = m->ipm[r + 4] = m->ipm[r + 5] = (int8_t)mode;
            }
        }
        c->chroma_mode = !d->asps->chroma_format_idc ? 0 /* no intra_chroma_pred_mode: DC */ : (cabac ? cabac_intra_chroma_mode(d) : (int)h264o_ue(b));
        if (c->chroma_mode > 3) return h264o_fail(d, "mb %d: bad intra_chroma_pred_mode", addr);
        m->chroma_mode = (uint8_t)c->chroma_mode;
    }
    if (MB_IS_INTER(c->type)) set_refids(d, m);

    if (c->type != MBT_I16x16) {
        int cbp;
        if (cabac)
            cbp = cabac_cbp(d);
        else {
            uint32_t k = h264o_ue(b);
            if (k > (d->asps->chroma_format_idc ? 47u : 15u)) return h264o_fail(d, "mb %d: bad coded_block_pattern", addr);
            if (d->asps->chroma_format_idc)
                cbp = MB_IS_INTRA(c->type) ? h264o_me_intra[k] : h264o_me_inter[k];
            else
                cbp = MB_IS_INTRA(c->type) ? h264o_me_intra0[k] : h264o_me_inter0[k];
        }
        c->cbp_luma = cbp & 15;
        c->cbp_chroma = cbp >> 4;
        if (c->cbp_luma && d->apps->transform_8x8_mode_flag && MB_IS_INTER(c->type)) {
            if (no_sub8) {
                c->t8x8 = cabac ? cabac_transform8x8(d) : (int)h264o_u(b, 1);
                m->t8x8 = (uint8_t)c->t8x8;
            }
        }
    }
    m->cbp_luma = (uint8_t)c->cbp_luma;
    m->cbp_chroma = (uint8_t)c->cbp_chroma;
    if (c->cbp_luma || c->cbp_chroma || c->type == MBT_I16x16) {
        int dqp = cabac ? cabac_mb_qp_delta(d) : h264o_se(b);
        if (dqp < -26 || dqp > 25) return h264o_fail(d, "mb %d: mb_qp_delta %d out of range", addr, dqp);
        d->prev_dqp_nz = dqp != 0;
        d->qp = (d->qp + dqp + 52) % 52;
        int r = parse_residual(d);
        if (r < 0) return r;
    } else
        d->prev_dqp_nz = 0;
    finish_mb(d);
    return b->err ? h264o_fail(d, "mb %d: bitstream overrun", addr) : 0;
}

/* ------------------------------------------------------------------ slice_data() 7.3.4 */
/* h264/slice.go:570-830.  Differences: mb_skip_flag / end_of_slice_flag are ae(v) (A23), mb_type
 * is tracked per MB (A22), residual() is parsed (A24), nextMbAddress follows the slice group map (8.2.2). */
int h264o_decode_slice_data(h264o_decoder *d) {
    h264o_br *b = &d->br;
    int cabac = d->apps->entropy_coding_mode_flag;
    int islice = d->sh.slice_type == 2;
    int bslice = d->sh.slice_type == 1;
    int total = d->wmb * d->hmb;
    int addr = d->sh.first_mb_in_slice;
    int64_t start_bits = b->pos;
    d->qp = d->sh.slice_qp_y;
    d->prev_dqp_nz = 0;
    if (cabac) {
        while (b->pos & 7)
            if (!h264o_u(b, 1)) return h264o_fail(d, "cabac_alignment_one_bit is 0");
        cabac_init_contexts(d);
        h264o_cabac_init_engine(d);
    }
    int more = 1;
    while (more) {
        if (addr >= total) return h264o_fail(d, "slice runs past the picture (mb %d)", addr);
        if (!islice) {
            if (!cabac) {
                uint32_t run = h264o_ue(b);
                if (run > (uint32_t)(total - addr)) return h264o_fail(d, "mb_skip_run %u too long", run);
                for (uint32_t i = 0; i < run; i++) {
                    if (addr >= total) return h264o_fail(d, "mb_skip_run runs past the slice group");
                    if ((bslice ? decode_bskip(d, addr) : decode_pskip(d, addr)) < 0) return -1;
                    addr = d->sgmap ? h264o_next_mb_address(d->sgmap, total, addr) : addr + 1; /* nextMbAddress, h264/slice.go:530 */
                }
                if (run > 0) more = h264o_more_rbsp_data(b);
                if (!more) break;
                if (addr >= total) return h264o_fail(d, "slice runs past the picture after skip run");
            } else {
                begin_mb(d, addr); /* neighbour helpers need c.mbx/mby */
                if (cabac_mb_skip_flag(d)) {
                    if ((bslice ? decode_bskip(d, addr) : decode_pskip(d, addr)) < 0) return -1;
                    goto end_mb;
                }
            }
        }
        {
            int r = decode_mb(d, addr);
            if (r < 0) return r;
        }
    end_mb:
        if (!cabac)
            more = h264o_more_rbsp_data(b);
        else
            more = !cabac_terminate(d);
        addr = d->sgmap ? h264o_next_mb_address(d->sgmap, total, addr) : addr + 1;
        if (b->err) return h264o_fail(d, "slice data overrun at mb %d", addr);
    }
    d->info.n_bits += (uint64_t)(b->pos - start_bits);
    return addr; /* one past the last MB of the slice */
}

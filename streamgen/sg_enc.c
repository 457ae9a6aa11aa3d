/*
 * streamgen/sg_enc.c -- synthetic-stream generator: source pictures, seeded mode decisions,
 * closed-loop reconstruction (sg_recon.c) and syntax writing (7.3: SPS, PPS, slice header,
 * slice_data, macroblock_layer, mb_pred, sub_mb_pred, residual) with CAVLC (9.2) or CABAC
 * (9.3.4) entropy coding.
 *
 * Input synthesis for tests/ and bench.py -- not part of the decode product, not the oracle.
 */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sg_int.h"

enum { T_NONE = 0, T_I4, T_I8, T_I16, T_PCM, T_P16, T_P16x8, T_P8x16, T_P8x8, T_SKIP, T_BDIRECT, T_BSKIP };
#define IS_SKIPPED(t) ((t) == T_SKIP || (t) == T_BSKIP)
#define IS_INTRA(t) ((t) >= T_I4 && (t) <= T_PCM)
#define IS_INTER(t) ((t) >= T_P16)

typedef struct {
    uint8_t type, t8x8, qp, qpc[2], cbp_luma, cbp_chroma, chroma_mode, i16mode, cbf_dc, dqp_nz;
    uint16_t slice_id, nzmask;
    int8_t ipm[16], ref[4];
    uint8_t nnz[24];
    int16_t mv[16][2], mvd[16][2];
    int32_t refid[4];
    uint8_t sub[4];
    /* list 1 of B macroblocks (ref / ref1 = -1: list not used by that 8x8 quadrant) */
    int8_t ref1[4];
    int16_t mv1[16][2], mvd1[16][2];
    int32_t refid1[4];
    uint8_t direct8; /* quadrants predicted in direct mode */
} emb;

typedef struct {
    sg_params p;
    int wmb, hmb, W, H;
    sg_pic pics[6];
    /* field_pics: the two fields of every frame store as pictures of their own (a frame coded as two fields is woven into pics[i]
     * afterwards, a frame coded as a frame is split into fpics[i][]), so that either kind of picture finds its references ready */
    sg_pic fpics[6][2];
    sg_pic *cur_frame;     /* the frame store the current picture belongs to (== cur for a frame picture) */
    int field, bottom, fH; /* coding a field picture, its parity; the coded FRAME height (W x H is the current PICTURE: frame or field) */
    sg_pic *cur, *refs[4];
    sg_pic *refs1[4];             /* RefPicList1 of a B picture */
    int nref1_active, cur_poc;
    uint16_t done_l[2];           /* B macroblocks: per list, 4x4 blocks whose motion is final */
    int cur_sub;                  /* sub-macroblock being coded: later ones are not available as neighbours (6.4.11.7) */
    int wb_w1[4], wb_o1[4], wb_cw1[4][2], wb_co1[4][2]; /* explicit weights of list 1 */
    int nrefs, next_id;
    emb *mb;
    sg_dbmb *db;
    uint8_t *src;
    uint64_t rng;
    sg_bw bw;
    int qp, prev_dqp_nz, slice_id, slice_type, skip_run, init_idc, nref_active;
    int ls4[6][6][16], ls8[2][6][64];
    uint8_t s4[6][16], s8[2][64]; /* scaling lists, zig-zag */
    int wp_w[4], wp_o[4], wp_cw[4][2], wp_co[4][2], wp_ld, wp_cd;
    /* picture management of the current picture (8.2.4.3 / 8.2.5.4), planned before its slice headers are written */
    int nal_ref_idc, cur_frame_num, n_rplm, n_mmco, max_lt, idr_lt, slice_qp, delta_poc0;
    struct { int idc, val; } rplm[8];
    struct { int op, a1, a2; } mmco[12];
    /* current MB */
    int mbx, mby, addr, raw_type;
    uint16_t done;
    int16_t i16dc[16], luma[16][16], luma8[4][64], cdc[2][4], cac[2][4][16];
    uint8_t pcm[384];
    /* slice groups (sg_params::slice_groups): PPS contents and the macroblock-to-slice-group map of the current picture */
    uint8_t *sgmap, *sg_ids;
    int sg_rl[8], sg_tl[8], sg_br[8], sg_dir, sg_rate, sg_cycle, sg_cycle_bits;
} enc;

static char g_err[256];
const char *sg_last_error(void) { return g_err; }

void sg_default_params(sg_params *p) {
    memset(p, 0, sizeof(*p));
    p->width = 176, p->height = 144, p->frames = 2, p->profile_idc = 66, p->qp = 28, p->idr_period = 1, p->slices = 1;
    p->num_ref_frames = 1, p->cabac_init_idc = 0, p->noise = 8, p->seed = 1, p->long_start_code = 1;
    p->intra_in_p_permille = 50, p->skip_permille = 250, p->sub8x8_permille = 100;
    p->motion_x4 = 12, p->motion_y4 = -8;
}

/* ------------------------------------------------------------------ PRNG + source */
static uint32_t rnd(enc *e) { /* xorshift64* */
    e->rng ^= e->rng >> 12;
    e->rng ^= e->rng << 25;
    e->rng ^= e->rng >> 27;
    return (uint32_t)((e->rng * 2685821657736338717ull) >> 32);
}
static int rnd_range(enc *e, int lo, int hi) { return lo + (int)(rnd(e) % (uint32_t)(hi - lo + 1)); }
static uint32_t hash32(uint32_t a) {
    a ^= a >> 16, a *= 0x7feb352du, a ^= a >> 15, a *= 0x846ca68bu, a ^= a >> 16;
    return a;
}
/* Y(x,y,t) = clip(128 + 64 sin((x+mx t)/37) + 48 cos((y+my t)/23) + n), n ~ U[-noise,noise]; chroma analogous.
 * The picture -- its noise field included -- moves (mx, my) = (motion_x4, motion_y4) / 4 samples per frame ((3, -2) by default),
 * so that motion compensation is meaningful; at fractional positions the noise field is interpolated bilinearly. */
static double noise_at(double xs, double ys, uint32_t mulx, uint32_t muly, uint32_t seed, uint32_t range, int bias) {
    double fx = floor(xs), fy = floor(ys), ax = xs - fx, ay = ys - fy, acc = 0;
    int xi = (int)fx, yi = (int)fy;
    for (int dy = 0; dy < 2; dy++)
        for (int dx = 0; dx < 2; dx++) {
            double w = (dx ? ax : 1 - ax) * (dy ? ay : 1 - ay);
            if (w == 0) continue;
            acc += w * ((int)(hash32(((uint32_t)(xi + dx) * mulx + (uint32_t)(yi + dy) * muly) ^ seed) % range) - bias);
        }
    return acc;
}
void sg_source_frame(const sg_params *p, int t, uint8_t *dst) {
    int W = (p->width + 15) & ~15, H = (p->height + 15) & ~15;
    uint8_t *y = dst, *cb = dst + W * H, *cr = cb + W * H / 4;
    const double mx = p->motion_x4 * t / 4.0, my = p->motion_y4 * t / 4.0;
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            double xs = i + mx, ys = j + my;
            double v = 128 + 64 * sin(xs / 37.0) + 48 * cos(ys / 23.0);
            int n = p->noise ? (int)floor(noise_at(xs, ys, 7919u, 104729u, p->seed, (uint32_t)(2 * p->noise + 1), p->noise) + 0.5) : 0;
            int q = (int)floor(v + 0.5) + n;
            y[j * W + i] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
        }
    for (int j = 0; j < H / 2; j++)
        for (int i = 0; i < W / 2; i++) {
            double xs = 2 * i + mx, ys = 2 * j + my;
            int n = p->noise ? (int)floor(noise_at(xs, ys, 31337u, 15485863u, p->seed * 3u, (uint32_t)(p->noise + 1), p->noise / 2) + 0.5) : 0;
            int a = (int)floor(128 + 40 * sin(xs / 53.0 + 1.0) + 24 * cos(ys / 41.0) + 0.5) + n;
            int b = (int)floor(128 + 36 * cos(xs / 47.0) + 30 * sin(ys / 29.0 + 2.0) + 0.5) - n;
            cb[j * (W / 2) + i] = (uint8_t)(a < 0 ? 0 : (a > 255 ? 255 : a));
            cr[j * (W / 2) + i] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
        }
    if (p->mono) memset(cb, 128, (size_t)(W * H / 2)); /* monochrome: the chroma a decoder puts out, so that prediction and residual of the (uncoded) planes are trivially 128 and 0 */
}

/* ------------------------------------------------------------------ helpers */
static int qpc_of(int qpi) {
    qpi = qpi < 0 ? 0 : (qpi > 51 ? 51 : qpi);
    return qpi < 30 ? qpi : sg_qpc_tab[qpi - 30];
}
static int blk_raster(int idx) { return (((idx >> 1) & 1) + 2 * (idx >> 3)) * 4 + ((idx & 1) + 2 * ((idx >> 2) & 1)); }
static emb *mb_at(enc *e, int mx, int my) {
    if (mx < 0 || my < 0 || mx >= e->wmb || my >= e->hmb) return NULL;
    emb *m = &e->mb[my * e->wmb + mx];
    if (m->type == T_NONE || m->slice_id != e->slice_id) return NULL;
    return m;
}
#define CURMB(e) (&(e)->mb[(e)->addr])
#define MBA(e) mb_at(e, (e)->mbx - 1, (e)->mby)
#define MBB(e) mb_at(e, (e)->mbx, (e)->mby - 1)
static emb *lnb(enc *e, int bx, int by, int *ri) {
    emb *m;
    if (bx < 0)
        m = MBA(e), bx += 4;
    else if (by < 0)
        m = MBB(e), by += 4;
    else
        m = CURMB(e);
    *ri = by * 4 + bx;
    return m;
}
static emb *cnb(enc *e, int cx, int cy, int *ri) {
    emb *m;
    if (cx < 0)
        m = MBA(e), cx += 2;
    else if (cy < 0)
        m = MBB(e), cy += 2;
    else
        m = CURMB(e);
    *ri = cy * 2 + cx;
    return m;
}
static int intra_ok(enc *e, int mx, int my) {
    emb *m = mb_at(e, mx, my);
    if (!m || my * e->wmb + mx >= e->addr) return 0;
    if (e->p.constrained_intra && IS_INTER(m->type)) return 0;
    return 1;
}

static void build_scale(enc *e) {
    for (int l = 0; l < 6; l++)
        for (int q = 0; q < 6; q++)
            for (int k = 0; k < 16; k++) {
                int r = sg_zigzag4x4[k], x = r & 3, y = r >> 2;
                int v = (!(x & 1) && !(y & 1)) ? sg_norm4x4[q][0] : (((x & 1) && (y & 1)) ? sg_norm4x4[q][1] : sg_norm4x4[q][2]);
                e->ls4[l][q][r] = e->s4[l][k] * v;
            }
    for (int l = 0; l < 2; l++)
        for (int q = 0; q < 6; q++)
            for (int k = 0; k < 64; k++) {
                int r = sg_zigzag8x8[k], x = r & 7, y = r >> 3, c;
                if (!(x & 3) && !(y & 3))
                    c = 0;
                else if ((x & 1) && (y & 1))
                    c = 1;
                else if ((x & 3) == 2 && (y & 3) == 2)
                    c = 2;
                else if ((!(y & 3) && (x & 1)) || ((y & 1) && !(x & 3)))
                    c = 3;
                else if ((!(y & 3) && (x & 3) == 2) || ((y & 3) == 2 && !(x & 3)))
                    c = 4;
                else
                    c = 5;
                e->ls8[l][q][r] = e->s8[l][k] * sg_norm8x8[q][c];
            }
}

/* ------------------------------------------------------------------ CAVLC writers (9.2) */
static void cavlc_block(enc *e, const int16_t *coef, int maxnum, int nC) {
    sg_bw *w = &e->bw;
    int level[16], run[16], total = 0, t1s = 0, last = -1;
    /* collect non-zero coefficients from high to low frequency */
    for (int i = maxnum - 1; i >= 0; i--)
        if (coef[i]) {
            level[total] = coef[i];
            if (last >= 0) run[total - 1] = last - i - 1;
            last = i;
            total++;
        }
    if (total) run[total - 1] = last; /* zeros before the lowest coefficient */
    for (int i = 0; i < total && i < 3; i++) {
        if (abs(level[i]) != 1) break;
        t1s++;
    }
    if (nC == -1)
        sg_put(w, sg_chroma_dc_token_bits[4 * total + t1s], sg_chroma_dc_token_len[4 * total + t1s]);
    else {
        int tbl = nC < 2 ? 0 : (nC < 4 ? 1 : (nC < 8 ? 2 : 3));
        sg_put(w, sg_coeff_token_bits[tbl][4 * total + t1s], sg_coeff_token_len[tbl][4 * total + t1s]);
    }
    if (!total) return;
    int suffix_len = (total > 10 && t1s < 3) ? 1 : 0;
    for (int i = 0; i < total; i++) {
        if (i < t1s) {
            sg_put(w, level[i] < 0, 1);
            continue;
        }
        int v = level[i];
        int code = v > 0 ? 2 * v - 2 : -2 * v - 1; /* levelCode */
        if (i == t1s && t1s < 3) code -= 2;
        /* 9.2.2.1 inverse */
        int prefix, suffix = 0, size = 0;
        if (suffix_len == 0) {
            if (code < 14)
                prefix = code;
            else if (code < 30)
                prefix = 14, size = 4, suffix = code - 14;
            else
                prefix = 15, size = 12, suffix = code - 30;
        } else {
            if (code < (15 << suffix_len))
                prefix = code >> suffix_len, size = suffix_len, suffix = code & ((1 << suffix_len) - 1);
            else
                prefix = 15, size = 12, suffix = code - (15 << suffix_len);
        }
        if (prefix == 15 && suffix >= 4096) { /* would need level_prefix >= 16: clamp (never hit at sane QPs) */
            suffix = 4095;
        }
        sg_put(w, 1, prefix + 1);
        if (size) sg_put(w, (uint32_t)suffix, size);
        if (suffix_len == 0) suffix_len = 1;
        if (abs(v) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }
    int total_zeros = 0;
    for (int i = 0; i < total; i++) total_zeros += run[i];
    if (total < maxnum) {
        if (maxnum == 4)
            sg_put(w, sg_chroma_dc_total_zeros_bits[total - 1][total_zeros], sg_chroma_dc_total_zeros_len[total - 1][total_zeros]);
        else
            sg_put(w, sg_total_zeros_bits[total - 1][total_zeros], sg_total_zeros_len[total - 1][total_zeros]);
    }
    int zl = total_zeros;
    for (int i = 0; i < total - 1 && zl > 0; i++) {
        int t = (zl > 7 ? 7 : zl) - 1;
        sg_put(w, sg_run_bits[t][run[i]], sg_run_len[t][run[i]]);
        zl -= run[i];
    }
}
static int count_nz(const int16_t *c, int n) {
    int k = 0;
    for (int i = 0; i < n; i++) k += c[i] != 0;
    return k;
}
static int nc_luma(enc *e, int bx, int by) {
    int ia, ib;
    emb *a = lnb(e, bx - 1, by, &ia), *b = lnb(e, bx, by - 1, &ib);
    if (a && b) return (a->nnz[ia] + b->nnz[ib] + 1) >> 1;
    return a ? a->nnz[ia] : (b ? b->nnz[ib] : 0);
}
static int nc_chroma(enc *e, int c, int cx, int cy) {
    int ia, ib;
    emb *a = cnb(e, cx - 1, cy, &ia), *b = cnb(e, cx, cy - 1, &ib);
    if (a && b) return (a->nnz[16 + 4 * c + ia] + b->nnz[16 + 4 * c + ib] + 1) >> 1;
    return a ? a->nnz[16 + 4 * c + ia] : (b ? b->nnz[16 + 4 * c + ib] : 0);
}

/* ------------------------------------------------------------------ CABAC writers (9.3.2 binarisations) */
static const int sig_off[5] = {0, 15, 29, 44, 47}, abs_off[5] = {0, 10, 20, 30, 39};
static void cabac_block(enc *e, const int16_t *coef, int cat, int maxnum, int cbf_inc) {
    sg_bw *w = &e->bw;
    int n = count_nz(coef, maxnum);
    if (cbf_inc >= 0) {
        sg_cabac_bin(w, 85 + cat * 4 + cbf_inc, n != 0);
        if (!n) return;
    }
    int lastpos = maxnum - 1;
    while (lastpos > 0 && !coef[lastpos]) lastpos--;
    for (int i = 0; i < maxnum - 1; i++) {
        int sctx, lctx;
        /* field pictures: the significance contexts of field-coded blocks (ctxIdxOffset 277 / 338, 8x8 blocks 436 / 451 with the field column of Table 9-43) */
        if (cat == 5)
            sctx = e->field ? 436 + sg_sig8x8_field_ctx[i] : 402 + sg_sig8x8_ctx[i], lctx = (e->field ? 451 : 417) + sg_last8x8_ctx[i];
        else {
            int inc = cat == 3 ? (i < 2 ? i : 2) : i;
            sctx = (e->field ? 277 : 105) + sig_off[cat] + inc, lctx = (e->field ? 338 : 166) + sig_off[cat] + inc;
        }
        sg_cabac_bin(w, sctx, coef[i] != 0);
        if (coef[i]) {
            sg_cabac_bin(w, lctx, i == lastpos);
            if (i == lastpos) break;
        }
    }
    int eq1 = 0, gt1 = 0, base = cat == 5 ? 426 : 227 + abs_off[cat];
    for (int i = lastpos; i >= 0; i--) {
        if (!coef[i]) continue;
        int a = abs(coef[i]) - 1; /* coeff_abs_level_minus1 */
        int inc0 = gt1 ? 0 : (1 + eq1 < 4 ? 1 + eq1 : 4);
        sg_cabac_bin(w, base + inc0, a > 0);
        if (a == 0)
            eq1++;
        else {
            int lim = 4 - (cat == 3), inc = 5 + (gt1 < lim ? gt1 : lim);
            int pre = a < 14 ? a : 14;
            for (int k = 1; k < pre; k++) sg_cabac_bin(w, base + inc, 1);
            if (a < 14)
                sg_cabac_bin(w, base + inc, 0);
            else { /* Exp-Golomb k=0 suffix of a-14 */
                int s = a - 14, k = 0;
                while (s >= (1 << k)) {
                    sg_cabac_bypass(w, 1);
                    s -= 1 << k;
                    k++;
                }
                sg_cabac_bypass(w, 0);
                while (k--) sg_cabac_bypass(w, (s >> k) & 1);
            }
            gt1++;
        }
        sg_cabac_bypass(w, coef[i] < 0);
    }
}
static int cbf_inc2(enc *e, emb *a, int fa, emb *b, int fb) {
    int ci = IS_INTRA(CURMB(e)->type);
    return (a ? fa : ci) + 2 * (b ? fb : ci);
}
static int cbf_luma(enc *e, int bx, int by) {
    int ia, ib;
    emb *a = lnb(e, bx - 1, by, &ia), *b = lnb(e, bx, by - 1, &ib);
    return cbf_inc2(e, a, a ? a->nnz[ia] != 0 : 0, b, b ? b->nnz[ib] != 0 : 0);
}
static int cbf_cac(enc *e, int c, int cx, int cy) {
    int ia, ib;
    emb *a = cnb(e, cx - 1, cy, &ia), *b = cnb(e, cx, cy - 1, &ib);
    return cbf_inc2(e, a, a ? a->nnz[16 + 4 * c + ia] != 0 : 0, b, b ? b->nnz[16 + 4 * c + ib] != 0 : 0);
}
static int cbf_dc(enc *e, int bit) {
    emb *a = MBA(e), *b = MBB(e);
    return cbf_inc2(e, a, a ? (a->cbf_dc >> bit) & 1 : 0, b, b ? (b->cbf_dc >> bit) & 1 : 0);
}
static void cabac_mvd(enc *e, int comp, int bx, int by, int v) {
    sg_bw *w = &e->bw;
    int ia, ib;
    emb *a = lnb(e, bx - 1, by, &ia), *b = lnb(e, bx, by - 1, &ib);
    int sum = (a ? a->mvd[ia][comp] : 0) + (b ? b->mvd[ib][comp] : 0), base = comp ? 47 : 40, m = abs(v);
    sg_cabac_bin(w, base + (sum > 2) + (sum > 32), m != 0);
    if (!m) return;
    int ctx = base + 3;
    for (int k = 1; k < (m < 9 ? m : 9); k++) {
        sg_cabac_bin(w, ctx, 1);
        if (k < 4) ctx++;
    }
    if (m < 9)
        sg_cabac_bin(w, ctx, 0);
    else {
        int s = m - 9, k = 3;
        while (s >= (1 << k)) {
            sg_cabac_bypass(w, 1);
            s -= 1 << k;
            k++;
        }
        sg_cabac_bypass(w, 0);
        while (k--) sg_cabac_bypass(w, (s >> k) & 1);
    }
    sg_cabac_bypass(w, v < 0);
}
static void cabac_ref(enc *e, int bx, int by, int ref) {
    int ia, ib;
    emb *a = lnb(e, bx - 1, by, &ia), *b = lnb(e, bx, by - 1, &ib);
    int ra = a ? a->ref[(ia >> 3) * 2 + ((ia & 3) >> 1)] : 0, rb = b ? b->ref[(ib >> 3) * 2 + ((ib & 3) >> 1)] : 0;
    int ctx = (ra > 0) + 2 * (rb > 0);
    for (int k = 0; k < ref; k++) {
        sg_cabac_bin(&e->bw, 54 + ctx, 1);
        ctx = (ctx >> 2) + 4;
    }
    sg_cabac_bin(&e->bw, 54 + ctx, 0);
}
static void cabac_intra_type(enc *e, int base, int islice, int it) {
    sg_bw *w = &e->bw;
    if (islice) {
        emb *a = MBA(e), *b = MBB(e);
        int inc = (a && a->type != T_I4 && a->type != T_I8) + (b && b->type != T_I4 && b->type != T_I8);
        sg_cabac_bin(w, base + inc, it != 0);
        base += 2;
    } else
        sg_cabac_bin(w, base, it != 0);
    if (it == 0) return;
    sg_cabac_terminate(w, it == 25);
    if (it == 25) return;
    int t = it - 1, lum = t >= 12, cc = (t >> 2) % 3, pm = t & 3;
    sg_cabac_bin(w, base + 1, lum);
    sg_cabac_bin(w, base + 2, cc != 0);
    if (cc) sg_cabac_bin(w, base + 2 + islice, cc == 2);
    sg_cabac_bin(w, base + 3 + islice, pm >> 1);
    sg_cabac_bin(w, base + 3 + 2 * islice, pm & 1);
}
/* B slices: first bin of mb_type (ctxIdx 27 + neighbours that are neither B_Skip nor B_Direct_16x16), Table 9-37 (b) */
static void cabac_b_type_bin0(enc *e, int bin) {
    emb *a = MBA(e), *b = MBB(e);
    int inc = (a && a->type != T_BSKIP && a->type != T_BDIRECT) + (b && b->type != T_BSKIP && b->type != T_BDIRECT);
    sg_cabac_bin(&e->bw, 27 + inc, bin);
}
static void cabac_b_intra_prefix(enc *e) { /* "111101", then the intra suffix at ctxIdxOffset 32 */
    sg_bw *w = &e->bw;
    cabac_b_type_bin0(e, 1);
    sg_cabac_bin(w, 27 + 3, 1);
    sg_cabac_bin(w, 27 + 4, 1);
    sg_cabac_bin(w, 27 + 5, 1);
    sg_cabac_bin(w, 27 + 5, 0);
    sg_cabac_bin(w, 27 + 5, 1);
}
static void cabac_b_mb_type(enc *e, int t) { /* mb_type 0..22 of Table 7-14 */
    sg_bw *w = &e->bw;
    cabac_b_type_bin0(e, t != 0);
    if (t == 0) return;
    if (t <= 2) {
        sg_cabac_bin(w, 27 + 3, 0);
        sg_cabac_bin(w, 27 + 5, t - 1);
        return;
    }
    sg_cabac_bin(w, 27 + 3, 1);
    int bits, extra = -1;
    if (t <= 10)
        bits = t - 3;
    else if (t == 11)
        bits = 14;
    else if (t == 22)
        bits = 15;
    else
        bits = (t + 4) >> 1, extra = (t + 4) & 1; /* 12..21 -> 8..12 + one more bin */
    sg_cabac_bin(w, 27 + 4, (bits >> 3) & 1);
    sg_cabac_bin(w, 27 + 5, (bits >> 2) & 1);
    sg_cabac_bin(w, 27 + 5, (bits >> 1) & 1);
    sg_cabac_bin(w, 27 + 5, bits & 1);
    if (extra >= 0) sg_cabac_bin(w, 27 + 5, extra);
}
static void cabac_b_sub_type(enc *e, int st) { /* sub_mb_type 0..12 of Table 7-18, Table 9-38 (b) */
    sg_bw *w = &e->bw;
    sg_cabac_bin(w, 36, st != 0);
    if (st == 0) return;
    if (st <= 2) {
        sg_cabac_bin(w, 37, 0);
        sg_cabac_bin(w, 39, st - 1);
        return;
    }
    sg_cabac_bin(w, 37, 1);
    if (st <= 6) {
        sg_cabac_bin(w, 38, 0);
        sg_cabac_bin(w, 39, ((st - 3) >> 1) & 1);
        sg_cabac_bin(w, 39, (st - 3) & 1);
    } else if (st <= 10) {
        sg_cabac_bin(w, 38, 1);
        sg_cabac_bin(w, 39, 0);
        sg_cabac_bin(w, 39, ((st - 7) >> 1) & 1);
        sg_cabac_bin(w, 39, (st - 7) & 1);
    } else {
        sg_cabac_bin(w, 38, 1);
        sg_cabac_bin(w, 39, 1);
        sg_cabac_bin(w, 39, st - 11);
    }
}
static void cabac_cbp(enc *e, int cbp) {
    sg_bw *w = &e->bw;
    emb *a = MBA(e), *b = MBB(e);
    int ca_ = a ? (a->type == T_PCM ? 0x2F : (a->cbp_luma | (a->cbp_chroma << 4))) : 0x0F;
    int cb_ = b ? (b->type == T_PCM ? 0x2F : (b->cbp_luma | (b->cbp_chroma << 4))) : 0x0F;
    for (int b8 = 0; b8 < 4; b8++) {
        int la = (b8 & 1) ? (cbp >> (b8 - 1)) & 1 : (ca_ >> (b8 + 1)) & 1;
        int lb = (b8 & 2) ? (cbp >> (b8 - 2)) & 1 : (cb_ >> (b8 + 2)) & 1;
        sg_cabac_bin(w, 73 + (!la) + 2 * (!lb), (cbp >> b8) & 1);
    }
    if (e->p.mono) return; /* ChromaArrayType 0: the prefix only */
    int cc = cbp >> 4;
    int fa = a && (a->type == T_PCM || a->cbp_chroma), fb = b && (b->type == T_PCM || b->cbp_chroma);
    sg_cabac_bin(w, 77 + fa + 2 * fb, cc != 0);
    if (cc) {
        fa = a && (a->type == T_PCM || a->cbp_chroma == 2), fb = b && (b->type == T_PCM || b->cbp_chroma == 2);
        sg_cabac_bin(w, 77 + 4 + fa + 2 * fb, cc == 2);
    }
}
static void cabac_dqp(enc *e, int dqp) {
    int val = dqp > 0 ? 2 * dqp - 1 : -2 * dqp, ctx = e->prev_dqp_nz ? 1 : 0;
    for (int k = 0; k < val; k++) {
        sg_cabac_bin(&e->bw, 60 + ctx, 1);
        ctx = 2 + (ctx >> 1);
    }
    sg_cabac_bin(&e->bw, 60 + ctx, 0);
}

/* ------------------------------------------------------------------ motion vector prediction (8.4.1.3) */
typedef struct {
    int ok, ref, x, y;
} nmv;
static nmv get_nmv(enc *e, int bx, int by) {
    nmv r = {0, -1, 0, 0};
    emb *m;
    if (by >= 0 && bx > 3) return r;
    if (bx >= 0 && bx <= 3 && by >= 0) {
        if (!(e->done >> (by * 4 + bx) & 1)) return r;
        m = CURMB(e);
    } else {
        m = mb_at(e, e->mbx + (bx < 0 ? -1 : (bx > 3 ? 1 : 0)), e->mby + (by < 0 ? -1 : 0));
        if (!m) return r;
        bx &= 3, by &= 3;
    }
    r.ok = 1;
    if (IS_INTRA(m->type)) return r;
    r.ref = m->ref[(by >> 1) * 2 + (bx >> 1)];
    r.x = m->mv[by * 4 + bx][0];
    r.y = m->mv[by * 4 + bx][1];
    return r;
}
static int mid3(int a, int b, int c) { return a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b)); }
/* shape: 0 median, 1/2 = 16x8 upper/lower, 3/4 = 8x16 left/right */
static void mv_pred(enc *e, int bx, int by, int w, int ref, int shape, int out[2]) {
    nmv A = get_nmv(e, bx - 1, by), B = get_nmv(e, bx, by - 1), C = get_nmv(e, bx + w, by - 1);
    if (!C.ok) C = get_nmv(e, bx - 1, by - 1);
    if (shape == 1 && B.ref == ref) { out[0] = B.x, out[1] = B.y; return; }
    if ((shape == 2 || shape == 3) && A.ref == ref) { out[0] = A.x, out[1] = A.y; return; }
    if (shape == 4 && C.ref == ref) { out[0] = C.x, out[1] = C.y; return; }
    if (!B.ok && !C.ok && A.ok) B = A, C = A;
    int hits = (A.ref == ref) + (B.ref == ref) + (C.ref == ref);
    if (hits == 1) {
        nmv s = A.ref == ref ? A : (B.ref == ref ? B : C);
        out[0] = s.x, out[1] = s.y;
        return;
    }
    out[0] = mid3(A.x, B.x, C.x);
    out[1] = mid3(A.y, B.y, C.y);
}
static void skip_mv(enc *e, int out[2]) {
    nmv A = get_nmv(e, -1, 0), B = get_nmv(e, 0, -1);
    out[0] = out[1] = 0;
    if (!A.ok || !B.ok) return;
    if ((A.ref == 0 && !A.x && !A.y) || (B.ref == 0 && !B.x && !B.y)) return;
    mv_pred(e, 0, 0, 4, 0, 0, out);
}
static void fill_part(enc *e, int bx, int by, int w, int h, const int mv[2], const int mvd[2]) {
    emb *m = CURMB(e);
    for (int y = by; y < by + h; y++)
        for (int x = bx; x < bx + w; x++) {
            m->mv[y * 4 + x][0] = (int16_t)mv[0], m->mv[y * 4 + x][1] = (int16_t)mv[1];
            m->mvd[y * 4 + x][0] = (int16_t)abs(mvd[0]), m->mvd[y * 4 + x][1] = (int16_t)abs(mvd[1]);
            e->done |= (uint16_t)(1 << (y * 4 + x));
        }
}

/* ------------------------------------------------------------------ prediction + residual coding of one MB */
static int sad(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h) {
    int s = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) s += abs(a[y * as + x] - b[y * bs + x]);
    return s;
}
static void put_block(uint8_t *dst, int stride, const uint8_t *pred, int n, const int *res) {
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            int v = pred[y * n + x] + (res ? res[y * n + x] : 0);
            dst[y * stride + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
}
static sg_avail mb_avail(enc *e) {
    sg_avail a;
    a.left = intra_ok(e, e->mbx - 1, e->mby);
    a.top = intra_ok(e, e->mbx, e->mby - 1);
    a.topleft = intra_ok(e, e->mbx - 1, e->mby - 1);
    a.topright = intra_ok(e, e->mbx + 1, e->mby - 1);
    return a;
}
static sg_avail blk4_avail(enc *e, int bx, int by) {
    sg_avail m = mb_avail(e), a;
    a.left = bx > 0 || m.left;
    a.top = by > 0 || m.top;
    a.topleft = (bx > 0 && by > 0) ? 1 : (bx > 0 ? m.top : (by > 0 ? m.left : m.topleft));
    if (by == 0)
        a.topright = bx < 3 ? m.top : m.topright;
    else if (bx == 3)
        a.topright = 0;
    else { /* inside the MB: the upper-right neighbour must precede in decoding order */
        static const uint8_t order[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15}; /* raster -> z index */
        a.topright = order[(by - 1) * 4 + bx + 1] < order[by * 4 + bx];
    }
    return a;
}
static sg_avail blk8_avail(enc *e, int b8) {
    sg_avail m = mb_avail(e), a;
    int x8 = b8 & 1, y8 = b8 >> 1;
    a.left = x8 || m.left;
    a.top = y8 || m.top;
    a.topleft = b8 == 0 ? m.topleft : (b8 == 1 ? m.top : (b8 == 2 ? m.left : 1));
    a.topright = b8 == 0 ? m.top : (b8 == 1 ? m.topright : (b8 == 2 ? 1 : 0));
    return a;
}
static int pred_ipm(enc *e, int bx, int by) {
    int ia, ib;
    emb *a = lnb(e, bx - 1, by, &ia), *b = lnb(e, bx, by - 1, &ib);
    if (!a || !b) return 2;
    if (e->p.constrained_intra && (IS_INTER(a->type) || IS_INTER(b->type))) return 2;
    int ma = (a->type == T_I4 || a->type == T_I8) ? a->ipm[ia] : 2, mb_ = (b->type == T_I4 || b->type == T_I8) ? b->ipm[ib] : 2;
    return ma < mb_ ? ma : mb_;
}
static double deadzone(int intra) { return intra ? 1.0 / 3 : 1.0 / 6; }

static void code_chroma(enc *e, int intra, const uint8_t pred[2][64]) {
    emb *m = CURMB(e);
    int W2 = e->W / 2, any_dc = 0, any_ac = 0;
    for (int c = 0; c < 2; c++) {
        const uint8_t *src = e->src + e->W * e->H + c * (e->W * e->H / 4) + e->mby * 8 * W2 + e->mbx * 8;
        int qp = m->qpc[c], list = (intra ? 1 : 4) + c;
        const int *ls = e->ls4[list][qp % 6];
        int sums[4];
        for (int b = 0; b < 4; b++) {
            int res[16], xo = (b & 1) * 4, yo = (b >> 1) * 4;
            sums[b] = 0;
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) {
                    res[y * 4 + x] = src[(yo + y) * W2 + xo + x] - pred[c][(yo + y) * 8 + xo + x];
                    sums[b] += res[y * 4 + x];
                }
            e->cac[c][b][0] = 0;
            int16_t lev[16];
            sg_quant4(res, ls, qp, deadzone(intra), 1, lev);
            memcpy(e->cac[c][b], lev, sizeof(lev));
            if (count_nz(lev, 16)) any_ac = 1;
        }
        sg_quant_chroma_dc(sums, ls[0], qp, deadzone(intra), e->cdc[c]);
        if (count_nz(e->cdc[c], 4)) any_dc = 1;
    }
    m->cbp_chroma = any_ac ? 2 : (any_dc ? 1 : 0);
    for (int c = 0; c < 2; c++) {
        uint8_t *dst = e->cur->pl[1 + c] + e->mby * 8 * W2 + e->mbx * 8;
        int qp = m->qpc[c], list = (intra ? 1 : 4) + c, dc[4];
        const int *ls = e->ls4[list][qp % 6];
        if (m->cbp_chroma == 0) memset(e->cdc[c], 0, sizeof(e->cdc[c]));
        if (m->cbp_chroma != 2) memset(e->cac[c], 0, sizeof(e->cac[c]));
        sg_chroma_dc(e->cdc[c], ls[0], qp, dc);
        for (int b = 0; b < 4; b++) {
            int res[16], xo = (b & 1) * 4, yo = (b >> 1) * 4;
            uint8_t pb[16];
            for (int y = 0; y < 4; y++) memcpy(pb + 4 * y, pred[c] + (yo + y) * 8 + xo, 4);
            if (m->cbp_chroma) {
                sg_residual4(e->cac[c][b], ls, qp, 1, dc[b], res);
                put_block(dst + yo * W2 + xo, W2, pb, 4, res);
            } else
                put_block(dst + yo * W2 + xo, W2, pb, 4, NULL);
        }
    }
}

/* luma residual for a predicted (non-I4/I8/I16) MB: inter */
static void code_luma_inter(enc *e, const uint8_t *pred /*16x16*/) {
    emb *m = CURMB(e);
    const uint8_t *src = e->src + e->mby * 16 * e->W + e->mbx * 16;
    uint8_t *dst = e->cur->pl[0] + e->mby * 16 * e->W + e->mbx * 16;
    int qp = m->qp;
    m->cbp_luma = 0;
    for (int b8 = 0; b8 < 4; b8++) {
        int x8 = (b8 & 1) * 8, y8 = (b8 >> 1) * 8;
        if (m->t8x8) {
            int res[64];
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) res[y * 8 + x] = src[(y8 + y) * e->W + x8 + x] - pred[(y8 + y) * 16 + x8 + x];
            sg_quant8(res, e->ls8[1][qp % 6], qp, deadzone(0), e->luma8[b8]);
            if (count_nz(e->luma8[b8], 64)) m->cbp_luma |= 1 << b8;
        } else
            for (int b4 = 0; b4 < 4; b4++) {
                int idx = b8 * 4 + b4, r = blk_raster(idx), xo = (r & 3) * 4, yo = (r >> 2) * 4, res[16];
                for (int y = 0; y < 4; y++)
                    for (int x = 0; x < 4; x++) res[y * 4 + x] = src[(yo + y) * e->W + xo + x] - pred[(yo + y) * 16 + xo + x];
                sg_quant4(res, e->ls4[3][qp % 6], qp, deadzone(0), 0, e->luma[idx]);
                if (count_nz(e->luma[idx], 16)) m->cbp_luma |= 1 << b8;
            }
    }
    for (int b8 = 0; b8 < 4; b8++) {
        int x8 = (b8 & 1) * 8, y8 = (b8 >> 1) * 8, coded = (m->cbp_luma >> b8) & 1;
        if (m->t8x8) {
            int res[64];
            uint8_t pb[64];
            for (int y = 0; y < 8; y++) memcpy(pb + 8 * y, pred + (y8 + y) * 16 + x8, 8);
            if (coded) sg_residual8(e->luma8[b8], e->ls8[1][qp % 6], qp, res);
            put_block(dst + y8 * e->W + x8, e->W, pb, 8, coded ? res : NULL);
        } else
            for (int b4 = 0; b4 < 4; b4++) {
                int idx = b8 * 4 + b4, r = blk_raster(idx), xo = (r & 3) * 4, yo = (r >> 2) * 4, res[16];
                uint8_t pb[16];
                for (int y = 0; y < 4; y++) memcpy(pb + 4 * y, pred + (yo + y) * 16 + xo, 4);
                if (coded) sg_residual4(e->luma[idx], e->ls4[3][qp % 6], qp, 0, 0, res);
                put_block(dst + yo * e->W + xo, e->W, pb, 4, coded ? res : NULL);
            }
    }
}

/* ------------------------------------------------------------------ syntax: residual() */
static void write_residual(enc *e) {
    emb *m = CURMB(e);
    int cabac = e->p.cabac, i16 = m->type == T_I16;
    if (i16) {
        int n = count_nz(e->i16dc, 16);
        if (cabac)
            cabac_block(e, e->i16dc, 0, 16, cbf_dc(e, 0));
        else
            cavlc_block(e, e->i16dc, 16, nc_luma(e, 0, 0));
        if (n) m->cbf_dc |= 1;
    }
    for (int b8 = 0; b8 < 4; b8++) {
        if (!(m->cbp_luma >> b8 & 1)) continue;
        int bx0 = (b8 & 1) * 2, by0 = (b8 >> 1) * 2;
        if (m->t8x8 && cabac) {
            int n = count_nz(e->luma8[b8], 64);
            cabac_block(e, e->luma8[b8], 5, 64, -1);
            m->nnz[by0 * 4 + bx0] = m->nnz[by0 * 4 + bx0 + 1] = m->nnz[by0 * 4 + 4 + bx0] = m->nnz[by0 * 4 + 5 + bx0] = (uint8_t)n;
            if (n) m->nzmask |= (uint16_t)(0x33 << (by0 * 4 + bx0));
            continue;
        }
        int any = 0;
        for (int b4 = 0; b4 < 4; b4++) {
            int idx = b8 * 4 + b4, r = blk_raster(idx), bx = r & 3, by = r >> 2, n;
            int16_t tmp[16];
            const int16_t *c = e->luma[idx];
            if (m->t8x8) { /* CAVLC 8x8: 4x4 "block" b4 holds coefficients 4*i + b4 of the 8x8 scan */
                for (int i = 0; i < 16; i++) tmp[i] = e->luma8[b8][4 * i + b4];
                c = tmp;
            }
            if (i16) {
                n = count_nz(c + 1, 15);
                if (cabac)
                    cabac_block(e, c + 1, 1, 15, cbf_luma(e, bx, by));
                else
                    cavlc_block(e, c + 1, 15, nc_luma(e, bx, by));
            } else {
                n = count_nz(c, 16);
                if (cabac)
                    cabac_block(e, c, 2, 16, cbf_luma(e, bx, by));
                else
                    cavlc_block(e, c, 16, nc_luma(e, bx, by));
            }
            m->nnz[r] = (uint8_t)n;
            if (n) m->nzmask |= (uint16_t)(1 << r), any = 1;
        }
        if (m->t8x8 && any) m->nzmask |= (uint16_t)(0x33 << (by0 * 4 + bx0));
    }
    if (m->cbp_chroma) {
        for (int c = 0; c < 2; c++) {
            int n = count_nz(e->cdc[c], 4);
            if (cabac)
                cabac_block(e, e->cdc[c], 3, 4, cbf_dc(e, 1 + c));
            else
                cavlc_block(e, e->cdc[c], 4, -1);
            if (n) m->cbf_dc |= (uint8_t)(2 << c);
        }
    }
    if (m->cbp_chroma & 2)
        for (int c = 0; c < 2; c++)
            for (int b = 0; b < 4; b++) {
                int n = count_nz(e->cac[c][b] + 1, 15);
                if (cabac)
                    cabac_block(e, e->cac[c][b] + 1, 4, 15, cbf_cac(e, c, b & 1, b >> 1));
                else
                    cavlc_block(e, e->cac[c][b] + 1, 15, nc_chroma(e, c, b & 1, b >> 1));
                m->nnz[16 + 4 * c + b] = (uint8_t)n;
            }
}

/* ------------------------------------------------------------------ MB begin/end */
static void begin_mb(enc *e, int addr) {
    emb *m = &e->mb[addr];
    e->addr = addr, e->mbx = addr % e->wmb, e->mby = addr / e->wmb, e->done = 0;
    memset(m, 0, sizeof(*m));
    m->slice_id = (uint16_t)e->slice_id;
    memset(m->ipm, -1, sizeof(m->ipm));
    memset(m->ref, -1, sizeof(m->ref));
    memset(m->ref1, -1, sizeof(m->ref1));
    for (int i = 0; i < 4; i++) m->refid[i] = m->refid1[i] = -1;
    e->done_l[0] = e->done_l[1] = 0, e->cur_sub = 3;
    memset(e->i16dc, 0, sizeof(e->i16dc));
    memset(e->luma, 0, sizeof(e->luma));
    memset(e->luma8, 0, sizeof(e->luma8));
    memset(e->cdc, 0, sizeof(e->cdc));
    memset(e->cac, 0, sizeof(e->cac));
}
static void set_qpc(enc *e, emb *m, int qp) {
    int second = e->p.chroma_qp_offset + ((e->p.profile_idc == 100 && e->p.transform8x8 && e->p.chroma_qp_offset < 12) ? 1 : 0); /* (7.4.2.2: -12 .. 12) */
    m->qp = (uint8_t)qp;
    m->qpc[0] = (uint8_t)qpc_of(qp + e->p.chroma_qp_offset);
    m->qpc[1] = (uint8_t)qpc_of(qp + second);
}
static void end_mb(enc *e) {
    emb *m = CURMB(e);
    sg_dbmb *d = &e->db[e->addr];
    d->intra = IS_INTRA(m->type);
    d->t8x8 = m->t8x8;
    if (m->type == T_PCM) {
        emb tmp;
        set_qpc(e, &tmp, 0);
        d->qp = 0, d->qpc[0] = tmp.qpc[0], d->qpc[1] = tmp.qpc[1];
    } else
        d->qp = m->qp, d->qpc[0] = m->qpc[0], d->qpc[1] = m->qpc[1];
    d->dbf_idc = (uint8_t)e->p.deblock_idc;
    d->alpha_off = (int8_t)(2 * e->p.alpha_off_div2);
    d->beta_off = (int8_t)(2 * e->p.beta_off_div2);
    d->slice_id = m->slice_id;
    d->nzmask = m->nzmask;
    memcpy(d->mv, m->mv, sizeof(d->mv));
    memcpy(d->refid, m->refid, sizeof(d->refid));
    memcpy(d->mv1, m->mv1, sizeof(d->mv1));
    memcpy(d->refid1, m->refid1, sizeof(d->refid1));
    for (int i = 0; i < 4; i++) { /* a list that is not used carries no picture */
        if (m->ref[i] < 0) d->refid[i] = -1;
        if (m->ref1[i] < 0) d->refid1[i] = -1;
    }
}

/* choose the QP this MB would like to use */
static int want_qp(enc *e) {
    if (!e->p.qp_jitter || rnd(e) % 8) return e->qp;
    int q = e->p.qp + rnd_range(e, -e->p.qp_jitter, e->p.qp_jitter);
    q = q < 10 ? 10 : (q > 51 ? 51 : q);
    if (q - e->qp > 25) q = e->qp + 25;
    if (q - e->qp < -26) q = e->qp - 26;
    return q;
}

/* ------------------------------------------------------------------ intra MB */
static void encode_intra(enc *e, int islice) {
    emb *m = CURMB(e);
    sg_bw *w = &e->bw;
    int cabac = e->p.cabac, W = e->W;
    const uint8_t *src = e->src + e->mby * 16 * W + e->mbx * 16;
    uint8_t *dst = e->cur->pl[0] + e->mby * 16 * W + e->mbx * 16;
    int r = rnd(e) % 1000, kind;
    if (r < e->p.pcm_permille)
        kind = T_PCM;
    else {
        int k = rnd(e) % 100;
        kind = e->p.transform8x8 ? (k < 35 ? T_I16 : (k < 70 ? T_I4 : T_I8)) : (k < 50 ? T_I16 : T_I4);
    }
    m->type = (uint8_t)kind;
    int qp = want_qp(e);
    set_qpc(e, m, qp);
    if (kind == T_PCM) {
        /* raw samples */
        for (int y = 0; y < 16; y++) memcpy(e->pcm + 16 * y, src + y * W, 16);
        for (int c = 0; c < 2; c++)
            for (int y = 0; y < 8; y++) memcpy(e->pcm + 256 + 64 * c + 8 * y, e->src + W * e->H + c * (W * e->H / 4) + (e->mby * 8 + y) * (W / 2) + e->mbx * 8, 8);
        for (int y = 0; y < 16; y++) memcpy(dst + y * W, e->pcm + 16 * y, 16);
        for (int c = 0; c < 2; c++)
            for (int y = 0; y < 8; y++) memcpy(e->cur->pl[1 + c] + (e->mby * 8 + y) * (W / 2) + e->mbx * 8, e->pcm + 256 + 64 * c + 8 * y, 8);
        if (cabac) {
            if (islice)
                cabac_intra_type(e, 3, 1, 25);
            else if (e->slice_type == 1) {
                cabac_b_intra_prefix(e);
                cabac_intra_type(e, 32, 0, 25);
            } else {
                sg_cabac_bin(w, 14, 1);
                cabac_intra_type(e, 17, 0, 25);
            }
        } else
            sg_put_ue(w, islice ? 25 : (e->slice_type == 1 ? 48 : 30));
        while (!sg_bw_aligned(w)) sg_put(w, 0, 1);
        for (int i = 0; i < (e->p.mono ? 256 : 384); i++) sg_put(w, e->pcm[i], 8);
        if (cabac) sg_cabac_start(w);
        set_qpc(e, m, e->qp); /* QP_Y unchanged across I_PCM */
        memset(m->nnz, 16, sizeof(m->nnz));
        m->nzmask = 0xFFFF, m->cbf_dc = 7, m->cbp_luma = 15, m->cbp_chroma = 2;
        e->prev_dqp_nz = 0;
        return;
    }
    sg_avail ma = mb_avail(e);
    /* ---- luma ---- */
    if (kind == T_I16) {
        uint8_t pred[256], best[256];
        int bs = 1 << 30, bm = 2;
        for (int mode = 0; mode < 4; mode++) {
            if (!sg_intra_mode_allowed(16, mode, &ma)) continue;
            sg_pred_i16(e->cur, e->mbx * 16, e->mby * 16, mode, &ma, pred);
            int s = sad(src, W, pred, 16, 16, 16) + (int)(rnd(e) % 64);
            if (s < bs) bs = s, bm = mode, memcpy(best, pred, 256);
        }
        m->i16mode = (uint8_t)bm;
        int sums[16], any_ac = 0;
        for (int idx = 0; idx < 16; idx++) {
            int rr = blk_raster(idx), xo = (rr & 3) * 4, yo = (rr >> 2) * 4, res[16];
            sums[rr] = 0;
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) res[y * 4 + x] = src[(yo + y) * W + xo + x] - best[(yo + y) * 16 + xo + x], sums[rr] += res[y * 4 + x];
            sg_quant4(res, e->ls4[0][qp % 6], qp, deadzone(1), 1, e->luma[idx]);
            if (count_nz(e->luma[idx], 16)) any_ac = 1;
        }
        sg_quant_luma_dc(sums, e->ls4[0][qp % 6][0], qp, deadzone(1), e->i16dc);
        m->cbp_luma = any_ac ? 15 : 0;
        if (!any_ac) memset(e->luma, 0, sizeof(e->luma));
        int dc[16];
        sg_luma_dc(e->i16dc, e->ls4[0][qp % 6][0], qp, dc);
        for (int idx = 0; idx < 16; idx++) {
            int rr = blk_raster(idx), xo = (rr & 3) * 4, yo = (rr >> 2) * 4, res[16];
            uint8_t pb[16];
            for (int y = 0; y < 4; y++) memcpy(pb + 4 * y, best + (yo + y) * 16 + xo, 4);
            sg_residual4(e->luma[idx], e->ls4[0][qp % 6], qp, 1, dc[rr], res);
            put_block(dst + yo * W + xo, W, pb, 4, res);
        }
    } else if (kind == T_I4) {
        for (int idx = 0; idx < 16; idx++) {
            int rr = blk_raster(idx), bx = rr & 3, by = rr >> 2, xo = bx * 4, yo = by * 4, res[16];
            sg_avail a = blk4_avail(e, bx, by);
            uint8_t pred[16], best[16];
            int bs = 1 << 30, bm = 2;
            for (int mode = 0; mode < 9; mode++) {
                if (!sg_intra_mode_allowed(4, mode, &a)) continue;
                sg_pred_i4(e->cur, e->mbx * 16 + xo, e->mby * 16 + yo, mode, &a, pred);
                int s = sad(src + yo * W + xo, W, pred, 4, 4, 4) + (int)(rnd(e) % 24);
                if (s < bs) bs = s, bm = mode, memcpy(best, pred, 16);
            }
            m->ipm[rr] = (int8_t)bm;
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) res[y * 4 + x] = src[(yo + y) * W + xo + x] - best[y * 4 + x];
            sg_quant4(res, e->ls4[0][qp % 6], qp, deadzone(1), 0, e->luma[idx]);
            /* cbp is per 8x8: decide after the 4th block of each 8x8; reconstruct assuming "coded" and
             * fix up below if the whole 8x8 quantised to zero (then residual is zero anyway) */
            sg_residual4(e->luma[idx], e->ls4[0][qp % 6], qp, 0, 0, res);
            put_block(dst + yo * W + xo, W, best, 4, res);
            if (count_nz(e->luma[idx], 16)) m->cbp_luma |= (uint8_t)(1 << (idx >> 2));
        }
    } else { /* T_I8 */
        m->t8x8 = 1;
        for (int b8 = 0; b8 < 4; b8++) {
            int xo = (b8 & 1) * 8, yo = (b8 >> 1) * 8, res[64];
            sg_avail a = blk8_avail(e, b8);
            uint8_t pred[64], best[64];
            int bs = 1 << 30, bm = 2;
            for (int mode = 0; mode < 9; mode++) {
                if (!sg_intra_mode_allowed(8, mode, &a)) continue;
                sg_pred_i8(e->cur, e->mbx * 16 + xo, e->mby * 16 + yo, mode, &a, pred);
                int s = sad(src + yo * W + xo, W, pred, 8, 8, 8) + (int)(rnd(e) % 48);
                if (s < bs) bs = s, bm = mode, memcpy(best, pred, 64);
            }
            int r0 = (b8 >> 1) * 8 + (b8 & 1) * 2;
            m->ipm[r0] = m->ipm[r0 + 1] = m->ipm[r0 + 4] = m->ipm[r0 + 5] = (int8_t)bm;
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) res[y * 8 + x] = src[(yo + y) * W + xo + x] - best[y * 8 + x];
            sg_quant8(res, e->ls8[0][qp % 6], qp, deadzone(1), e->luma8[b8]);
            if (count_nz(e->luma8[b8], 64)) {
                m->cbp_luma |= (uint8_t)(1 << b8);
                sg_residual8(e->luma8[b8], e->ls8[0][qp % 6], qp, res);
                put_block(dst + yo * W + xo, W, best, 8, res);
            } else
                put_block(dst + yo * W + xo, W, best, 8, NULL);
        }
    }
    /* ---- chroma ---- */
    {
        uint8_t pred[2][64], tmp[2][64];
        int bs = 1 << 30, bm = 0;
        for (int mode = 0; mode < (e->p.mono ? 1 : 4); mode++) { /* (monochrome: no intra_chroma_pred_mode -- DC of planes that are 128 everywhere) */
            if (!sg_intra_mode_allowed(0, mode, &ma)) continue;
            int s = (int)(rnd(e) % 32);
            for (int c = 0; c < 2; c++) {
                sg_pred_chroma(e->cur, 1 + c, e->mbx * 8, e->mby * 8, mode, &ma, tmp[c]);
                s += sad(e->src + W * e->H + c * (W * e->H / 4) + e->mby * 8 * (W / 2) + e->mbx * 8, W / 2, tmp[c], 8, 8, 8);
            }
            if (s < bs) bs = s, bm = mode, memcpy(pred, tmp, sizeof(pred));
        }
        m->chroma_mode = (uint8_t)bm;
        code_chroma(e, 1, pred);
    }
    /* ---- syntax ---- */
    int it = kind == T_I16 ? 1 + m->i16mode + 4 * m->cbp_chroma + (m->cbp_luma ? 12 : 0) : 0;
    e->raw_type = islice ? it : it + (e->slice_type == 1 ? 23 : 5);
    if (cabac) {
        if (islice)
            cabac_intra_type(e, 3, 1, it);
        else if (e->slice_type == 1) {
            cabac_b_intra_prefix(e);
            cabac_intra_type(e, 32, 0, it);
        } else {
            sg_cabac_bin(w, 14, 1);
            cabac_intra_type(e, 17, 0, it);
        }
    } else
        sg_put_ue(w, (uint32_t)e->raw_type);
    if (kind != T_I16 && e->p.transform8x8) {
        if (cabac) {
            emb *a = MBA(e), *b = MBB(e);
            sg_cabac_bin(w, 399 + (a && a->t8x8) + (b && b->t8x8), m->t8x8);
        } else
            sg_put(w, m->t8x8, 1);
    }
    if (kind != T_I16) {
        int n = kind == T_I8 ? 4 : 16;
        for (int i = 0; i < n; i++) {
            int rr = n == 4 ? (i >> 1) * 8 + (i & 1) * 2 : blk_raster(i), bx = rr & 3, by = rr >> 2;
            /* pred_ipm must only see modes of blocks that precede this one: ipm of later blocks is still
             * final here, but the derivation only looks left/up, which always precede. */
            int pred = pred_ipm(e, bx, by), mode = m->ipm[rr];
            if (mode == pred) {
                if (cabac)
                    sg_cabac_bin(w, 68, 1);
                else
                    sg_put(w, 1, 1);
            } else {
                int rem = mode < pred ? mode : mode - 1;
                if (cabac) {
                    sg_cabac_bin(w, 68, 0);
                    sg_cabac_bin(w, 69, rem & 1);
                    sg_cabac_bin(w, 69, (rem >> 1) & 1);
                    sg_cabac_bin(w, 69, (rem >> 2) & 1);
                } else {
                    sg_put(w, 0, 1);
                    sg_put(w, (uint32_t)rem, 3);
                }
            }
        }
    }
    if (e->p.mono)
        ; /* ChromaArrayType 0: no intra_chroma_pred_mode */
    else if (cabac) {
        emb *a = MBA(e), *b = MBB(e);
        int inc = (a && IS_INTRA(a->type) && a->type != T_PCM && a->chroma_mode) + (b && IS_INTRA(b->type) && b->type != T_PCM && b->chroma_mode);
        sg_cabac_bin(w, 64 + inc, m->chroma_mode != 0);
        if (m->chroma_mode) {
            sg_cabac_bin(w, 67, m->chroma_mode > 1);
            if (m->chroma_mode > 1) sg_cabac_bin(w, 67, m->chroma_mode > 2);
        }
    } else
        sg_put_ue(w, m->chroma_mode);
    if (kind != T_I16) {
        int cbp = m->cbp_luma | (m->cbp_chroma << 4);
        if (cabac)
            cabac_cbp(e, cbp);
        else {
            int k = 0;
            while ((e->p.mono ? sg_me_intra0[k] : sg_me_intra[k]) != cbp) k++;
            sg_put_ue(w, (uint32_t)k);
        }
    }
    if (m->cbp_luma || m->cbp_chroma || kind == T_I16) {
        int dqp = qp - e->qp;
        if (cabac)
            cabac_dqp(e, dqp);
        else
            sg_put_se(w, dqp);
        e->prev_dqp_nz = dqp != 0;
        e->qp = qp;
        write_residual(e);
    } else {
        e->prev_dqp_nz = 0;
        set_qpc(e, m, e->qp);
    }
}

/* ------------------------------------------------------------------ inter MB */
static void pick_mv(enc *e, int x, int y, int w, int h, sg_pic *ref, const int mvp[2], int out[2]) {
    const uint8_t *src = e->src + y * e->W + x;
    int cand[6][2], n = 0, best = 0, bs = 1 << 30;
    uint8_t tmp[256];
    cand[n][0] = mvp[0], cand[n++][1] = mvp[1];
    const int tmx = e->p.motion_x4, tmy = e->p.motion_y4;
    cand[n][0] = tmx, cand[n++][1] = tmy; /* true motion of the synthetic scene per frame */
    cand[n][0] = tmx + rnd_range(e, -6, 6), cand[n++][1] = tmy + rnd_range(e, -6, 6);
    cand[n][0] = tmx + rnd_range(e, -3, 3), cand[n++][1] = tmy + rnd_range(e, -3, 3);
    cand[n][0] = rnd_range(e, -64, 64), cand[n++][1] = rnd_range(e, -64, 64);
    cand[n][0] = 0, cand[n++][1] = 0;
    if (rnd(e) % 100 < 25)
        best = (int)(rnd(e) % (uint32_t)n);
    else
        for (int i = 0; i < n; i++) {
            sg_mc_luma(ref, x, y, w, h, cand[i][0], cand[i][1], tmp, 16);
            int s = sad(src, e->W, tmp, 16, w, h);
            if (s < bs) bs = s, best = i;
        }
    out[0] = cand[best][0], out[1] = cand[best][1];
    /* keep the displaced block within 32 samples of the picture so that streams stay level-conformant */
    int minx = -4 * (x + 24), maxx = 4 * (e->W - x - w + 24), miny = -4 * (y + 24), maxy = 4 * (e->H - y - h + 24);
    out[0] = out[0] < minx ? minx : (out[0] > maxx ? maxx : out[0]);
    out[1] = out[1] < miny ? miny : (out[1] > maxy ? maxy : out[1]);
}
static void mc_part(enc *e, int bx, int by, int w, int h, uint8_t *py, uint8_t pc[2][64]) {
    emb *m = CURMB(e);
    int ref = m->ref[(by >> 1) * 2 + (bx >> 1)];
    sg_pic *rp = e->refs[ref];
    int mvx = m->mv[by * 4 + bx][0], mvy = m->mv[by * 4 + bx][1];
    sg_mc_luma(rp, e->mbx * 16 + bx * 4, e->mby * 16 + by * 4, w * 4, h * 4, mvx, mvy, py + by * 4 * 16 + bx * 4, 16);
    /* Table 8-9: predicting from a field of the other parity shifts the chroma vector by a quarter chroma sample */
    const int cofs = !e->field || rp->parity == e->bottom ? 0 : (e->bottom ? 2 : -2);
    for (int c = 0; c < 2; c++) sg_mc_chroma(rp, 1 + c, e->mbx * 8 + bx * 2, e->mby * 8 + by * 2, w * 2, h * 2, mvx, mvy + cofs, pc[c] + by * 2 * 8 + bx * 2, 8);
    if (e->p.weighted_pred) {
        int ld = e->wp_ld, w0 = e->wp_w[ref], o0 = e->wp_o[ref];
        for (int y = by * 4; y < (by + h) * 4; y++)
            for (int x = bx * 4; x < (bx + w) * 4; x++) {
                int v = py[y * 16 + x];
                v = ld >= 1 ? ((v * w0 + (1 << (ld - 1))) >> ld) + o0 : v * w0 + o0;
                py[y * 16 + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        for (int c = 0; c < 2; c++) {
            int cd = e->wp_cd, cw = e->wp_cw[ref][c], co = e->wp_co[ref][c];
            for (int y = by * 2; y < (by + h) * 2; y++)
                for (int x = bx * 2; x < (bx + w) * 2; x++) {
                    int v = pc[c][y * 8 + x];
                    v = cd >= 1 ? ((v * cw + (1 << (cd - 1))) >> cd) + co : v * cw + co;
                    pc[c][y * 8 + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                }
        }
    }
}
typedef struct {
    int bx, by, w, h, shape, mvd[2];
} part;

static void set_refids(enc *e, emb *m) {
    for (int i = 0; i < 4; i++) m->refid[i] = m->ref[i] >= 0 ? e->refs[m->ref[i]]->id : -1;
}

static void encode_skip(enc *e) {
    emb *m = CURMB(e);
    int mv[2], zero[2] = {0, 0};
    uint8_t py[256], pc[2][64];
    m->type = T_SKIP;
    memset(m->ref, 0, sizeof(m->ref));
    skip_mv(e, mv);
    fill_part(e, 0, 0, 4, 4, mv, zero);
    set_refids(e, m);
    set_qpc(e, m, e->qp);
    mc_part(e, 0, 0, 4, 4, py, pc);
    for (int y = 0; y < 16; y++) memcpy(e->cur->pl[0] + (e->mby * 16 + y) * e->W + e->mbx * 16, py + 16 * y, 16);
    for (int c = 0; c < 2; c++)
        for (int y = 0; y < 8; y++) memcpy(e->cur->pl[1 + c] + (e->mby * 8 + y) * (e->W / 2) + e->mbx * 8, pc[c] + 8 * y, 8);
    e->prev_dqp_nz = 0;
    e->raw_type = -1;
}

static void encode_inter(enc *e, int kind) {
    emb *m = CURMB(e);
    sg_bw *w = &e->bw;
    int cabac = e->p.cabac, nref = e->nref_active;
    part parts[16];
    int np = 0;
    m->type = (uint8_t)kind;
    /* reference indices */
    int refs4[4];
    for (int i = 0; i < 4; i++) refs4[i] = (nref > 1 && rnd(e) % 100 < 35) ? rnd_range(e, 0, nref - 1) : 0;
    if (kind == T_P16)
        refs4[1] = refs4[2] = refs4[3] = refs4[0];
    else if (kind == T_P16x8)
        refs4[1] = refs4[0], refs4[3] = refs4[2];
    else if (kind == T_P8x16)
        refs4[2] = refs4[0], refs4[3] = refs4[1];
    for (int i = 0; i < 4; i++) m->ref[i] = (int8_t)refs4[i];
    set_refids(e, m);
    if (kind == T_P16)
        parts[np++] = (part){0, 0, 4, 4, 0, {0, 0}};
    else if (kind == T_P16x8) {
        parts[np++] = (part){0, 0, 4, 2, 1, {0, 0}};
        parts[np++] = (part){0, 2, 4, 2, 2, {0, 0}};
    } else if (kind == T_P8x16) {
        parts[np++] = (part){0, 0, 2, 4, 3, {0, 0}};
        parts[np++] = (part){2, 0, 2, 4, 4, {0, 0}};
    } else
        for (int i = 0; i < 4; i++) {
            int bx = (i & 1) * 2, by = (i >> 1) * 2, st = rnd(e) % 100 < 55 ? 0 : rnd_range(e, 1, 3);
            m->sub[i] = (uint8_t)st;
            if (st == 0)
                parts[np++] = (part){bx, by, 2, 2, 0, {0, 0}};
            else if (st == 1) {
                parts[np++] = (part){bx, by, 2, 1, 0, {0, 0}};
                parts[np++] = (part){bx, by + 1, 2, 1, 0, {0, 0}};
            } else if (st == 2) {
                parts[np++] = (part){bx, by, 1, 2, 0, {0, 0}};
                parts[np++] = (part){bx + 1, by, 1, 2, 0, {0, 0}};
            } else {
                parts[np++] = (part){bx, by, 1, 1, 0, {0, 0}};
                parts[np++] = (part){bx + 1, by, 1, 1, 0, {0, 0}};
                parts[np++] = (part){bx, by + 1, 1, 1, 0, {0, 0}};
                parts[np++] = (part){bx + 1, by + 1, 1, 1, 0, {0, 0}};
            }
        }
    /* motion per partition, in decoding order */
    uint8_t py[256], pc[2][64];
    for (int i = 0; i < np; i++) {
        part *p = &parts[i];
        int mvp[2], mv[2], ref = m->ref[(p->by >> 1) * 2 + (p->bx >> 1)];
        mv_pred(e, p->bx, p->by, p->w, ref, p->shape, mvp);
        pick_mv(e, e->mbx * 16 + p->bx * 4, e->mby * 16 + p->by * 4, p->w * 4, p->h * 4, e->refs[ref], mvp, mv);
        p->mvd[0] = mv[0] - mvp[0], p->mvd[1] = mv[1] - mvp[1];
        fill_part(e, p->bx, p->by, p->w, p->h, mv, p->mvd);
        mc_part(e, p->bx, p->by, p->w, p->h, py, pc);
    }
    int qp = want_qp(e);
    set_qpc(e, m, qp);
    int all8 = 1;
    if (kind == T_P8x8)
        for (int i = 0; i < 4; i++)
            if (m->sub[i]) all8 = 0;
    m->t8x8 = (e->p.transform8x8 && all8 && rnd(e) % 2) ? 1 : 0;
    code_luma_inter(e, py);
    if (!m->cbp_luma) m->t8x8 = 0;
    code_chroma(e, 0, pc);
    /* ---- syntax ---- */
    int raw = kind == T_P16 ? 0 : (kind == T_P16x8 ? 1 : (kind == T_P8x16 ? 2 : 3));
    e->raw_type = raw;
    if (cabac) {
        sg_cabac_bin(w, 14, 0);
        if (raw == 0 || raw == 3) {
            sg_cabac_bin(w, 15, 0);
            sg_cabac_bin(w, 16, raw == 3);
        } else {
            sg_cabac_bin(w, 15, 1);
            sg_cabac_bin(w, 17, raw == 1);
        }
    } else
        sg_put_ue(w, (uint32_t)raw);
    if (kind == T_P8x8)
        for (int i = 0; i < 4; i++) {
            if (cabac) {
                int st = m->sub[i];
                sg_cabac_bin(w, 21, st == 0);
                if (st) {
                    sg_cabac_bin(w, 22, st != 1);
                    if (st != 1) sg_cabac_bin(w, 23, st == 2);
                }
            } else
                sg_put_ue(w, m->sub[i]);
        }
    if (nref > 1) {
        /* the CABAC ctxIdxInc for ref_idx looks at refs of neighbouring partitions in *this* MB too: they
         * are already final in m->ref[], but partitions that come later must read as "not yet known" =
         * they are never consulted (A/B neighbours always precede). */
        int nparts_ref = kind == T_P16 ? 1 : (kind == T_P8x8 ? 4 : 2);
        for (int i = 0; i < nparts_ref; i++) {
            int bx, by;
            if (kind == T_P16)
                bx = by = 0;
            else if (kind == T_P16x8)
                bx = 0, by = i * 2;
            else if (kind == T_P8x16)
                bx = i * 2, by = 0;
            else
                bx = (i & 1) * 2, by = (i >> 1) * 2;
            int ref = m->ref[(by >> 1) * 2 + (bx >> 1)];
            if (cabac)
                cabac_ref(e, bx, by, ref);
            else
                sg_put_te(w, nref - 1, (uint32_t)ref);
        }
    }
    for (int i = 0; i < np; i++) {
        part *p = &parts[i];
        if (cabac) {
            cabac_mvd(e, 0, p->bx, p->by, p->mvd[0]);
            cabac_mvd(e, 1, p->bx, p->by, p->mvd[1]);
        } else {
            sg_put_se(w, p->mvd[0]);
            sg_put_se(w, p->mvd[1]);
        }
    }
    int cbp = m->cbp_luma | (m->cbp_chroma << 4);
    if (cabac)
        cabac_cbp(e, cbp);
    else {
        int k = 0;
        while ((e->p.mono ? sg_me_inter0[k] : sg_me_inter[k]) != cbp) k++;
        sg_put_ue(w, (uint32_t)k);
    }
    if (m->cbp_luma && e->p.transform8x8 && all8) {
        if (cabac) {
            emb *a = MBA(e), *b = MBB(e);
            sg_cabac_bin(w, 399 + (a && a->t8x8) + (b && b->t8x8), m->t8x8);
        } else
            sg_put(w, m->t8x8, 1);
    }
    if (cbp) {
        int dqp = qp - e->qp;
        if (cabac)
            cabac_dqp(e, dqp);
        else
            sg_put_se(w, dqp);
        e->prev_dqp_nz = dqp != 0;
        e->qp = qp;
        write_residual(e);
    } else {
        e->prev_dqp_nz = 0;
        set_qpc(e, m, e->qp);
    }
}


/* ------------------------------------------------------------------ B macroblocks (7.3.5 with Table 7-14 / 7-18, 8.4.1.2)
 * Written next to the P path, not into it: P streams stay bit-identical.  Motion is kept per list; a quadrant uses a
 * list when its ref / ref1 entry is >= 0. */
#define EMV(m, l) ((l) ? (m)->mv1 : (m)->mv)
#define EMVD(m, l) ((l) ? (m)->mvd1 : (m)->mvd)
#define EREF(m, l) ((l) ? (m)->ref1 : (m)->ref)
#define EREFID(m, l) ((l) ? (m)->refid1 : (m)->refid)

static nmv get_nmv_l(enc *e, int l, int bx, int by) {
    nmv r = {0, -1, 0, 0};
    emb *m;
    if (by >= 0 && bx > 3) return r;
    if (bx >= 0 && bx <= 3 && by >= 0) {
        if (!(e->done_l[l] >> (by * 4 + bx) & 1)) return r;
        if ((by >> 1) * 2 + (bx >> 1) > e->cur_sub) return r; /* a later sub-macroblock (possible for direct ones, derived up front) */
        m = CURMB(e);
    } else {
        m = mb_at(e, e->mbx + (bx < 0 ? -1 : (bx > 3 ? 1 : 0)), e->mby + (by < 0 ? -1 : 0));
        if (!m) return r;
        bx &= 3, by &= 3;
    }
    r.ok = 1;
    if (IS_INTRA(m->type)) return r;
    r.ref = EREF(m, l)[(by >> 1) * 2 + (bx >> 1)];
    if (r.ref < 0) return r;
    r.x = EMV(m, l)[by * 4 + bx][0];
    r.y = EMV(m, l)[by * 4 + bx][1];
    return r;
}
static void mv_pred_l(enc *e, int l, int bx, int by, int w, int ref, int shape, int out[2]) {
    nmv A = get_nmv_l(e, l, bx - 1, by), B = get_nmv_l(e, l, bx, by - 1), C = get_nmv_l(e, l, bx + w, by - 1);
    if (!C.ok) C = get_nmv_l(e, l, bx - 1, by - 1);
    if (shape == 1 && B.ref == ref) { out[0] = B.x, out[1] = B.y; return; }
    if ((shape == 2 || shape == 3) && A.ref == ref) { out[0] = A.x, out[1] = A.y; return; }
    if (shape == 4 && C.ref == ref) { out[0] = C.x, out[1] = C.y; return; }
    if (!B.ok && !C.ok && A.ok) B = A, C = A;
    int hits = (A.ref == ref) + (B.ref == ref) + (C.ref == ref);
    if (hits == 1) {
        nmv s = A.ref == ref ? A : (B.ref == ref ? B : C);
        out[0] = s.x, out[1] = s.y;
        return;
    }
    out[0] = mid3(A.x, B.x, C.x);
    out[1] = mid3(A.y, B.y, C.y);
}
static void fill_part_l(enc *e, int l, int bx, int by, int w, int h, const int mv[2], const int mvd[2]) {
    emb *m = CURMB(e);
    for (int y = by; y < by + h; y++)
        for (int x = bx; x < bx + w; x++) {
            EMV(m, l)[y * 4 + x][0] = (int16_t)mv[0], EMV(m, l)[y * 4 + x][1] = (int16_t)mv[1];
            EMVD(m, l)[y * 4 + x][0] = (int16_t)abs(mvd[0]), EMVD(m, l)[y * 4 + x][1] = (int16_t)abs(mvd[1]);
            e->done_l[l] |= (uint16_t)(1 << (y * 4 + x));
        }
}
static void cabac_mvd_l(enc *e, int l, int comp, int bx, int by, int v) {
    sg_bw *w = &e->bw;
    int ia, ib;
    emb *a = lnb(e, bx - 1, by, &ia), *b = lnb(e, bx, by - 1, &ib);
    int sum = (a ? EMVD(a, l)[ia][comp] : 0) + (b ? EMVD(b, l)[ib][comp] : 0), base = comp ? 47 : 40, m = abs(v);
    sg_cabac_bin(w, base + (sum > 2) + (sum > 32), m != 0);
    if (!m) return;
    int ctx = base + 3;
    for (int k = 1; k < (m < 9 ? m : 9); k++) {
        sg_cabac_bin(w, ctx, 1);
        if (k < 4) ctx++;
    }
    if (m < 9)
        sg_cabac_bin(w, ctx, 0);
    else {
        int s = m - 9, k = 3;
        while (s >= (1 << k)) {
            sg_cabac_bypass(w, 1);
            s -= 1 << k;
            k++;
        }
        sg_cabac_bypass(w, 0);
        while (k--) sg_cabac_bypass(w, (s >> k) & 1);
    }
    sg_cabac_bypass(w, v < 0);
}
static void cabac_ref_l(enc *e, int l, int bx, int by, int ref) {
    int ia, ib;
    emb *a = lnb(e, bx - 1, by, &ia), *b = lnb(e, bx, by - 1, &ib);
    int qa = (ia >> 3) * 2 + ((ia & 3) >> 1), qb = (ib >> 3) * 2 + ((ib & 3) >> 1);
    /* quadrants predicted in direct mode count as refIdx 0 here (9.3.3.1.1.6) */
    int ra = (a && !(a->direct8 >> qa & 1)) ? EREF(a, l)[qa] : 0, rb = (b && !(b->direct8 >> qb & 1)) ? EREF(b, l)[qb] : 0;
    int ctx = (ra > 0) + 2 * (rb > 0);
    for (int k = 0; k < ref; k++) {
        sg_cabac_bin(&e->bw, 54 + ctx, 1);
        ctx = (ctx >> 2) + 4;
    }
    sg_cabac_bin(&e->bw, 54 + ctx, 0);
}

/* co-located block of (bx, by) in RefPicList1[0] (direct_8x8_inference_flag = 1: the corner block of the quadrant) */
static int col_block(enc *e, int bx, int by, int mv[2], int *pic_id) {
    const sg_pic *col = e->refs1[0];
    const emb *cm = col && col->motion ? &((const emb *)col->motion)[e->addr] : NULL;
    mv[0] = mv[1] = 0, *pic_id = -1;
    if (!cm || !IS_INTER(cm->type)) return -1;
    int cx = (bx >> 1) * 3, cy = (by >> 1) * 3, q = (cy >> 1) * 2 + (cx >> 1), l = cm->ref[q] >= 0 ? 0 : 1;
    if (EREF(cm, l)[q] < 0) return -1;
    mv[0] = EMV(cm, l)[cy * 4 + cx][0], mv[1] = EMV(cm, l)[cy * 4 + cx][1];
    *pic_id = EREFID(cm, l)[q];
    return EREF(cm, l)[q];
}
static int poc_clip(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }
/* DistScaleFactor of 8.4.1.2.3 / 8.4.2.3.1; returns 0 when no scaling applies (long-term or equal POCs) */
static int dist_scale(int cur, int p0, int p1, int lt, int *dsf) {
    int tb = poc_clip(cur - p0), td = poc_clip(p1 - p0);
    if (lt || td == 0) return 0;
    int tx = (16384 + abs(td / 2)) / td, v = (tb * tx + 32) >> 6;
    *dsf = v < -1024 ? -1024 : (v > 1023 ? 1023 : v);
    return 1;
}
/* motion of the quadrants in mask8, predicted in direct mode */
static void b_direct(enc *e, int mask8) {
    emb *m = CURMB(e);
    const int zero[2] = {0, 0};
    if (!e->p.direct_temporal) { /* spatial: reference indices and vectors from the neighbours A, B, C of the macroblock */
        int ref[2], mvp[2][2] = {{0, 0}, {0, 0}};
        for (int l = 0; l < 2; l++) {
            nmv A = get_nmv_l(e, l, -1, 0), B = get_nmv_l(e, l, 0, -1), C = get_nmv_l(e, l, 4, -1);
            if (!C.ok) C = get_nmv_l(e, l, -1, -1);
            int r = -1; /* the smallest non-negative one */
            if (A.ref >= 0) r = A.ref;
            if (B.ref >= 0 && (r < 0 || B.ref < r)) r = B.ref;
            if (C.ref >= 0 && (r < 0 || C.ref < r)) r = C.ref;
            ref[l] = r;
        }
        if (ref[0] < 0 && ref[1] < 0)
            ref[0] = ref[1] = 0;
        else
            for (int l = 0; l < 2; l++)
                if (ref[l] >= 0) mv_pred_l(e, l, 0, 0, 4, ref[l], 0, mvp[l]);
        int col_is_short = e->refs1[0] && e->refs1[0]->is_ref == 1;
        for (int q = 0; q < 4; q++) {
            if (!(mask8 >> q & 1)) continue;
            int cmv[2], cid, cref = col_block(e, (q & 1) * 2, (q >> 1) * 2, cmv, &cid);
            int still = col_is_short && cref == 0 && abs(cmv[0]) <= 1 && abs(cmv[1]) <= 1; /* colZeroFlag */
            for (int l = 0; l < 2; l++) {
                EREF(m, l)[q] = (int8_t)ref[l];
                fill_part_l(e, l, (q & 1) * 2, (q >> 1) * 2, 2, 2, (ref[l] < 0 || (ref[l] == 0 && still)) ? zero : mvp[l], zero);
            }
        }
        return;
    }
    for (int q = 0; q < 4; q++) { /* temporal: the co-located vector scaled by the POC distances */
        if (!(mask8 >> q & 1)) continue;
        int cmv[2], cid, cref = col_block(e, (q & 1) * 2, (q >> 1) * 2, cmv, &cid), r0 = 0;
        if (cref >= 0) {
            r0 = -1;
            for (int i = 0; i < e->nref_active && r0 < 0; i++)
                if (e->refs[i]->id == cid) r0 = i;
            if (r0 < 0) { /* cannot happen: the anchors' references stay in the DPB and within the active entries (plan_ref_list, plan_field_list) */
                if (getenv("SG_DEBUG")) fprintf(stderr, "temporal direct: the co-located block's reference is not in RefPicList0\n");
                r0 = 0, cmv[0] = cmv[1] = 0;
            }
        }
        int dsf = 0, mv0[2], mv1[2];
        if (dist_scale(e->cur_poc, e->refs[r0]->poc, e->refs1[0]->poc, e->refs[r0]->is_ref == 2, &dsf)) {
            for (int c = 0; c < 2; c++) mv0[c] = (dsf * cmv[c] + 128) >> 8, mv1[c] = mv0[c] - cmv[c];
        } else
            mv0[0] = cmv[0], mv0[1] = cmv[1], mv1[0] = mv1[1] = 0;
        m->ref[q] = (int8_t)r0, m->ref1[q] = 0;
        fill_part_l(e, 0, (q & 1) * 2, (q >> 1) * 2, 2, 2, mv0, zero);
        fill_part_l(e, 1, (q & 1) * 2, (q >> 1) * 2, 2, 2, mv1, zero);
    }
}

/* prediction samples of the whole macroblock from its per-list motion (8.4.2.2, 8.4.2.3) */
static void b_predict(enc *e, uint8_t *py, uint8_t pc[2][64]) {
    emb *m = CURMB(e);
    const int idc = e->p.weighted_bipred;
    for (int blk = 0; blk < 16; blk++) {
        int bx = blk & 3, by = blk >> 2, q = (by >> 1) * 2 + (bx >> 1);
        uint8_t ty[2][16] = {{0}}, tc[2][2][4] = {{{0}}}; /* (a list that is not used is read below, its value dropped) */
        int use[2] = {m->ref[q] >= 0, m->ref1[q] >= 0};
        sg_pic *rp[2] = {use[0] ? e->refs[m->ref[q]] : NULL, use[1] ? e->refs1[m->ref1[q]] : NULL};
        for (int l = 0; l < 2; l++) {
            if (!use[l]) continue;
            int mvx = EMV(m, l)[blk][0], mvy = EMV(m, l)[blk][1];
            sg_mc_luma(rp[l], e->mbx * 16 + bx * 4, e->mby * 16 + by * 4, 4, 4, mvx, mvy, ty[l], 4);
            const int cofs = !e->field || rp[l]->parity == e->bottom ? 0 : (e->bottom ? 2 : -2); /* Table 8-9, as in mc_part() */
            for (int c = 0; c < 2; c++) sg_mc_chroma(rp[l], 1 + c, e->mbx * 8 + bx * 2, e->mby * 8 + by * 2, 2, 2, mvx, mvy + cofs, tc[l][c], 2);
        }
        int iw[2] = {32, 32};
        if (idc == 2 && use[0] && use[1]) {
            int dsf = 0;
            if (rp[1]->is_ref != 2 && dist_scale(e->cur_poc, rp[0]->poc, rp[1]->poc, rp[0]->is_ref == 2, &dsf) && (dsf >> 2) >= -64 && (dsf >> 2) <= 128)
                iw[0] = 64 - (dsf >> 2), iw[1] = dsf >> 2;
        }
        for (int comp = 0; comp < 3; comp++) {
            int n = comp ? 4 : 16, wd = comp ? 2 : 4;
            int ld = comp ? e->wp_cd : e->wp_ld;
            int w0 = use[0] ? (comp ? e->wp_cw[m->ref[q]][comp - 1] : e->wp_w[m->ref[q]]) : 0, o0 = use[0] ? (comp ? e->wp_co[m->ref[q]][comp - 1] : e->wp_o[m->ref[q]]) : 0;
            int w1 = use[1] ? (comp ? e->wb_cw1[m->ref1[q]][comp - 1] : e->wb_w1[m->ref1[q]]) : 0, o1 = use[1] ? (comp ? e->wb_co1[m->ref1[q]][comp - 1] : e->wb_o1[m->ref1[q]]) : 0;
            for (int i = 0; i < n; i++) {
                int a = comp ? tc[0][comp - 1][i] : ty[0][i], b = comp ? tc[1][comp - 1][i] : ty[1][i], v;
                if (use[0] && use[1]) {
                    if (idc == 1)
                        v = ((a * w0 + b * w1 + (1 << ld)) >> (ld + 1)) + ((o0 + o1 + 1) >> 1);
                    else if (idc == 2)
                        v = (a * iw[0] + b * iw[1] + 32) >> 6;
                    else
                        v = (a + b + 1) >> 1;
                } else {
                    v = use[0] ? a : b;
                    if (idc == 1) {
                        int ww = use[0] ? w0 : w1, oo = use[0] ? o0 : o1;
                        v = ld >= 1 ? ((v * ww + (1 << (ld - 1))) >> ld) + oo : v * ww + oo;
                    }
                }
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
                if (comp == 0)
                    py[(by * 4 + i / wd) * 16 + bx * 4 + i % wd] = (uint8_t)v;
                else
                    pc[comp - 1][(by * 2 + i / wd) * 8 + bx * 2 + i % wd] = (uint8_t)v;
            }
        }
    }
}

static const uint8_t b_pair_modes[9][2] = {{1, 1}, {2, 2}, {1, 2}, {2, 1}, {1, 3}, {2, 3}, {3, 1}, {3, 2}, {3, 3}}; /* Table 7-14, mb_type 4..21 */
static const uint8_t b_sub_mode[13] = {0, 1, 2, 3, 1, 1, 2, 2, 3, 3, 1, 2, 3};                                      /* Table 7-18: 0 = direct */
static const uint8_t b_sub_shape[13] = {0, 0, 0, 0, 1, 2, 1, 2, 1, 2, 3, 3, 3};                                     /* 0 8x8, 1 8x4, 2 4x8, 3 4x4 */
typedef struct {
    int bx, by, w, h, shape, mode, sub, mvd[2][2];
} bpart;

/* kind: T_BSKIP, T_BDIRECT, T_P16 (16x16), T_P16x8, T_P8x16, T_P8x8 */
static void encode_b(enc *e, int kind) {
    emb *m = CURMB(e);
    sg_bw *w = &e->bw;
    const int cabac = e->p.cabac;
    bpart pt[16];
    int np = 0, raw = 0;
    m->type = (uint8_t)kind;
    if (kind == T_BSKIP || kind == T_BDIRECT) {
        m->direct8 = 15;
        b_direct(e, 15);
    } else if (kind == T_P8x8) {
        raw = 22;
        for (int i = 0; i < 4; i++) {
            int st = (int)(rnd(e) % 13);
            m->sub[i] = (uint8_t)st;
            if (b_sub_mode[st] == 0) m->direct8 |= (uint8_t)(1 << i);
        }
        if (m->direct8) {
            e->cur_sub = -1;
            b_direct(e, m->direct8);
        }
        for (int i = 0; i < 4; i++) {
            int st = m->sub[i], bx = (i & 1) * 2, by = (i >> 1) * 2, md = b_sub_mode[st];
            if (!md) continue;
            switch (b_sub_shape[st]) {
            case 0: pt[np++] = (bpart){bx, by, 2, 2, 0, md, i, {{0, 0}, {0, 0}}}; break;
            case 1: pt[np++] = (bpart){bx, by, 2, 1, 0, md, i, {{0, 0}, {0, 0}}}, pt[np++] = (bpart){bx, by + 1, 2, 1, 0, md, i, {{0, 0}, {0, 0}}}; break;
            case 2: pt[np++] = (bpart){bx, by, 1, 2, 0, md, i, {{0, 0}, {0, 0}}}, pt[np++] = (bpart){bx + 1, by, 1, 2, 0, md, i, {{0, 0}, {0, 0}}}; break;
            default:
                for (int k = 0; k < 4; k++) pt[np++] = (bpart){bx + (k & 1), by + (k >> 1), 1, 1, 0, md, i, {{0, 0}, {0, 0}}};
            }
        }
    } else {
        int m0 = 1 + (int)(rnd(e) % 3), m1 = 1 + (int)(rnd(e) % 3);
        if (kind == T_P16) {
            raw = m0;
            pt[np++] = (bpart){0, 0, 4, 4, 0, m0, 3, {{0, 0}, {0, 0}}};
        } else {
            int k = 0;
            while (b_pair_modes[k][0] != m0 || b_pair_modes[k][1] != m1) k++;
            raw = 4 + 2 * k + (kind == T_P8x16);
            if (kind == T_P16x8)
                pt[np++] = (bpart){0, 0, 4, 2, 1, m0, 3, {{0, 0}, {0, 0}}}, pt[np++] = (bpart){0, 2, 4, 2, 2, m1, 3, {{0, 0}, {0, 0}}};
            else
                pt[np++] = (bpart){0, 0, 2, 4, 3, m0, 3, {{0, 0}, {0, 0}}}, pt[np++] = (bpart){2, 0, 2, 4, 4, m1, 3, {{0, 0}, {0, 0}}};
        }
    }
    /* reference indices: one per (quadrant-aligned partition, list); then the vectors, list by list in decoding order */
    const int nref[2] = {e->nref_active, e->nref1_active};
    for (int l = 0; l < 2; l++) {
        if (kind == T_P8x8) { /* one index per non-direct quadrant that uses the list */
            for (int i = 0; i < 4; i++)
                if (b_sub_mode[m->sub[i]] >> l & 1) EREF(m, l)[i] = (int8_t)((nref[l] > 1 && rnd(e) % 100 < 35) ? rnd_range(e, 0, nref[l] - 1) : 0);
            continue;
        }
        for (int i = 0; i < np; i++) {
            if (!(pt[i].mode >> l & 1)) continue;
            int r = (nref[l] > 1 && rnd(e) % 100 < 35) ? rnd_range(e, 0, nref[l] - 1) : 0;
            for (int y = pt[i].by >> 1; y < (pt[i].by + pt[i].h + 1) >> 1; y++)
                for (int x = pt[i].bx >> 1; x < (pt[i].bx + pt[i].w + 1) >> 1; x++) EREF(m, l)[y * 2 + x] = (int8_t)r;
        }
    }
    for (int l = 0; l < 2; l++)
        for (int i = 0; i < np; i++) {
            bpart *p = &pt[i];
            if (!(p->mode >> l & 1)) {
                /* the partition does not use this list: for the partitions after it, it is an AVAILABLE neighbour with
                 * refIdxLX = -1 and a zero vector (8.4.1.3.2) -- availability is a matter of decoding order (6.4.11.7) */
                for (int y = p->by; y < p->by + p->h; y++)
                    for (int x = p->bx; x < p->bx + p->w; x++) e->done_l[l] |= (uint16_t)(1 << (y * 4 + x));
                continue;
            }
            int mvp[2], mv[2], ref = EREF(m, l)[(p->by >> 1) * 2 + (p->bx >> 1)];
            e->cur_sub = p->sub;
            mv_pred_l(e, l, p->bx, p->by, p->w, ref, p->shape, mvp);
            pick_mv(e, e->mbx * 16 + p->bx * 4, e->mby * 16 + p->by * 4, p->w * 4, p->h * 4, l ? e->refs1[ref] : e->refs[ref], mvp, mv);
            p->mvd[l][0] = mv[0] - mvp[0], p->mvd[l][1] = mv[1] - mvp[1];
            fill_part_l(e, l, p->bx, p->by, p->w, p->h, mv, p->mvd[l]);
        }
    e->cur_sub = 3;
    for (int i = 0; i < 4; i++) {
        m->refid[i] = m->ref[i] >= 0 ? e->refs[m->ref[i]]->id : -1;
        m->refid1[i] = m->ref1[i] >= 0 ? e->refs1[m->ref1[i]]->id : -1;
    }
    uint8_t py[256], pc[2][64];
    b_predict(e, py, pc);
    if (kind == T_BSKIP) {
        set_qpc(e, m, e->qp);
        for (int y = 0; y < 16; y++) memcpy(e->cur->pl[0] + (e->mby * 16 + y) * e->W + e->mbx * 16, py + 16 * y, 16);
        for (int c = 0; c < 2; c++)
            for (int y = 0; y < 8; y++) memcpy(e->cur->pl[1 + c] + (e->mby * 8 + y) * (e->W / 2) + e->mbx * 8, pc[c] + 8 * y, 8);
        e->prev_dqp_nz = 0;
        e->raw_type = -1;
        return;
    }
    int qp = want_qp(e);
    set_qpc(e, m, qp);
    int all8 = 1; /* no sub-partition smaller than 8x8 (direct quadrants count as 8x8: direct_8x8_inference_flag = 1) */
    if (kind == T_P8x8)
        for (int i = 0; i < 4; i++)
            if (b_sub_shape[m->sub[i]]) all8 = 0;
    m->t8x8 = (e->p.transform8x8 && all8 && rnd(e) % 2) ? 1 : 0;
    code_luma_inter(e, py);
    if (!m->cbp_luma) m->t8x8 = 0;
    code_chroma(e, 0, pc);
    /* ---- syntax ---- */
    e->raw_type = raw;
    if (cabac)
        cabac_b_mb_type(e, raw);
    else
        sg_put_ue(w, (uint32_t)raw);
    if (kind == T_P8x8)
        for (int i = 0; i < 4; i++) {
            if (cabac)
                cabac_b_sub_type(e, m->sub[i]);
            else
                sg_put_ue(w, m->sub[i]);
        }
    for (int l = 0; l < 2; l++) {
        if (nref[l] <= 1) continue;
        if (kind == T_P8x8) {
            for (int i = 0; i < 4; i++) {
                if (!(b_sub_mode[m->sub[i]] >> l & 1)) continue;
                if (cabac)
                    cabac_ref_l(e, l, (i & 1) * 2, (i >> 1) * 2, EREF(m, l)[i]);
                else
                    sg_put_te(w, nref[l] - 1, (uint32_t)EREF(m, l)[i]);
            }
            continue;
        }
        for (int i = 0; i < np; i++) {
            if (!(pt[i].mode >> l & 1)) continue;
            int ref = EREF(m, l)[(pt[i].by >> 1) * 2 + (pt[i].bx >> 1)];
            if (cabac)
                cabac_ref_l(e, l, pt[i].bx, pt[i].by, ref);
            else
                sg_put_te(w, nref[l] - 1, (uint32_t)ref);
        }
    }
    for (int l = 0; l < 2; l++)
        for (int i = 0; i < np; i++) {
            if (!(pt[i].mode >> l & 1)) continue;
            if (cabac) {
                cabac_mvd_l(e, l, 0, pt[i].bx, pt[i].by, pt[i].mvd[l][0]);
                cabac_mvd_l(e, l, 1, pt[i].bx, pt[i].by, pt[i].mvd[l][1]);
            } else {
                sg_put_se(w, pt[i].mvd[l][0]);
                sg_put_se(w, pt[i].mvd[l][1]);
            }
        }
    int cbp = m->cbp_luma | (m->cbp_chroma << 4);
    if (cabac)
        cabac_cbp(e, cbp);
    else {
        int k = 0;
        while ((e->p.mono ? sg_me_inter0[k] : sg_me_inter[k]) != cbp) k++;
        sg_put_ue(w, (uint32_t)k);
    }
    if (m->cbp_luma && e->p.transform8x8 && all8) {
        if (cabac) {
            emb *a = MBA(e), *b = MBB(e);
            sg_cabac_bin(w, 399 + (a && a->t8x8) + (b && b->t8x8), m->t8x8);
        } else
            sg_put(w, m->t8x8, 1);
    }
    if (cbp) {
        int dqp = qp - e->qp;
        if (cabac)
            cabac_dqp(e, dqp);
        else
            sg_put_se(w, dqp);
        e->prev_dqp_nz = dqp != 0;
        e->qp = qp;
        write_residual(e);
    } else {
        e->prev_dqp_nz = 0;
        set_qpc(e, m, e->qp);
    }
}

/* The CABAC mvd ctxIdxInc of partition k reads |mvd| of neighbouring partitions that precede k.  In
 * encode_inter all partitions' mvd are already stored when the syntax is written, which would let
 * a later partition's values leak into an earlier one's context if the A/B neighbour were a LATER
 * partition -- impossible, since A (left) and B (above) always precede in decoding order. */

/* ------------------------------------------------------------------ headers */
static size_t write_sps(enc *e, uint8_t *dst, size_t cap) {
    uint8_t buf[512];
    sg_bw w;
    const sg_params *p = &e->p;
    sg_bw_init(&w, buf, sizeof(buf));
    sg_put(&w, (uint32_t)p->profile_idc, 8);
    sg_put(&w, p->profile_idc == 66 ? 0xC0 : (p->profile_idc == 77 ? 0x40 : 0), 8); /* constraint_set flags */
    sg_put(&w, e->W * e->fH > 1920 * 1088 ? 51 : 40, 8);                            /* level_idc */
    sg_put_ue(&w, 0);                                                              /* sps id */
    if (p->profile_idc == 100) {
        sg_put_ue(&w, p->mono ? 0 : 1); /* chroma_format_idc */
        sg_put_ue(&w, 0);
        sg_put_ue(&w, 0);
        sg_put(&w, 0, 1); /* qpprime_y_zero_transform_bypass */
        sg_put(&w, p->scaling_matrix ? 1 : 0, 1);
        if (p->scaling_matrix) {
            /* lists 0 and 3 and 6,7: useDefault (delta_scale -8 at j=0 -> nextScale 0); others fall back */
            for (int i = 0; i < 8; i++) {
                int send = (i == 0 || i == 3 || i == 6 || i == 7);
                sg_put(&w, (uint32_t)send, 1);
                if (send) sg_put_se(&w, -8);
            }
        }
    }
    sg_put_ue(&w, 4); /* log2_max_frame_num_minus4 -> 8 bits */
    sg_put_ue(&w, (uint32_t)p->poc_type);
    if (p->poc_type == 0) sg_put_ue(&w, 4); /* log2_max_poc_lsb_minus4 -> 8 bits */
    if (p->poc_type == 1) {
        sg_put(&w, 0, 1);   /* delta_pic_order_always_zero_flag */
        sg_put_se(&w, -1);  /* offset_for_non_ref_pic */
        sg_put_se(&w, 1);   /* offset_for_top_to_bottom_field: PicOrderCnt = min(top, top + 1) = top */
        sg_put_ue(&w, 2);   /* num_ref_frames_in_pic_order_cnt_cycle */
        sg_put_se(&w, 2);   /* offset_for_ref_frame[0] */
        sg_put_se(&w, 6);   /* offset_for_ref_frame[1] */
    }
    sg_put_ue(&w, (uint32_t)p->num_ref_frames);
    sg_put(&w, p->fn_gap_declared != 0, 1); /* gaps_in_frame_num_value_allowed_flag */
    sg_put_ue(&w, (uint32_t)(e->wmb - 1));
    if (p->interlace_sps) { /* map units of two macroblock rows; crop units of four luma rows (7.4.2.1.1) */
        sg_put_ue(&w, (uint32_t)(e->fH / 32 - 1));
        sg_put(&w, 0, 1); /* frame_mbs_only_flag */
        sg_put(&w, 0, 1); /* mb_adaptive_frame_field_flag */
    } else {
        sg_put_ue(&w, (uint32_t)(e->fH / 16 - 1));
        sg_put(&w, 1, 1); /* frame_mbs_only */
    }
    sg_put(&w, 1, 1); /* direct_8x8_inference */
    int cr = (e->W - p->width) / 2, cb = (e->fH - p->height) / (p->interlace_sps ? 4 : 2);
    sg_put(&w, cr || cb, 1);
    if (cr || cb) {
        sg_put_ue(&w, 0);
        sg_put_ue(&w, (uint32_t)cr);
        sg_put_ue(&w, 0);
        sg_put_ue(&w, (uint32_t)cb);
    }
    sg_put(&w, 0, 1); /* vui */
    sg_trailing(&w);
    return sg_write_nal(dst, cap, 1, 3, 7, buf, sg_bw_bytes(&w));
}
/* ------------------------------------------------------------------ slice groups (7.3.2.2, 8.2.2) */
static int sg_map_units(const enc *e) { return e->wmb * (e->p.interlace_sps ? e->fH / 32 : e->fH / 16); } /* (of the FRAME: the same for its frame and field pictures) */
/* what the PPS will say, chosen from the seed so that every map type gets awkward shapes */
static void plan_slice_groups(enc *e) {
    const sg_params *p = &e->p;
    int W = e->wmb, Hm = sg_map_units(e) / W, units = W * Hm, ng = p->slice_groups;
    for (int g = 0; g < 8; g++) {
        e->sg_rl[g] = (int)((p->seed * 5u + 7u * (unsigned)g) % (unsigned)(W + 3 < units ? W + 3 : units)); /* run_length_minus1: runs that straddle rows; 7.4.2.2: at most PicSizeInMapUnits - 1 */
        int x0 = (2 * g + 1) % (W > 2 ? W / 2 : 1), y0 = (g + 1) % (Hm > 2 ? Hm / 2 : 1);
        int x1 = x0 + W / 3, y1 = y0 + Hm / 3;
        if (x1 > W - 1) x1 = W - 1;
        if (y1 > Hm - 1) y1 = Hm - 1;
        e->sg_tl[g] = y0 * W + x0, e->sg_br[g] = y1 * W + x1;
    }
    e->sg_dir = (int)(p->seed & 1u);
    e->sg_rate = W / 2 + 1; /* SliceGroupChangeRate: half a row and a bit */
    e->sg_cycle_bits = 0;
    while (((1ll << e->sg_cycle_bits) - 1) * e->sg_rate < units) e->sg_cycle_bits++; /* Ceil(Log2(units / rate + 1)), exact quotient */
    e->sg_ids = (uint8_t *)malloc((size_t)units);
    uint32_t h = p->seed * 2654435761u + 12345u;
    for (int i = 0; i < units; i++) { /* explicit map: mostly the neighbour's group, sometimes a new one -- ragged blobs */
        h = h * 1664525u + 1013904223u;
        e->sg_ids[i] = (uint8_t)((i > 0 && (h >> 24) % 3 != 0) ? e->sg_ids[i - 1] : (h >> 16) % (unsigned)ng);
    }
    e->sgmap = (uint8_t *)malloc((size_t)e->wmb * e->hmb);
}
/* mapUnitToSliceGroupMap for slice_group_change_cycle `cycle`, then mbToSliceGroupMap (8.2.2.8) into e->sgmap */
static void build_slice_group_map(enc *e, int cycle) {
    const sg_params *p = &e->p;
    int W = e->wmb, Hm = sg_map_units(e) / W, units = W * Hm, ng = p->slice_groups;
    uint8_t *mu = (uint8_t *)malloc((size_t)units);
    int in0 = cycle * e->sg_rate < units ? cycle * e->sg_rate : units;
    int upper_left = e->sg_dir ? units - in0 : in0;
    switch (p->fmo_type) {
    case 0:
        for (int i = 0, g = 0; i < units; g = (g + 1) % ng)
            for (int j = 0; j <= e->sg_rl[g] && i < units; j++) mu[i++] = (uint8_t)g;
        break;
    case 1:
        for (int y = 0; y < Hm; y++)
            for (int x = 0; x < W; x++) mu[y * W + x] = (uint8_t)((x + ((y * ng) >> 1)) % ng);
        break;
    case 2:
        for (int i = 0; i < units; i++) {
            int x = i % W, y = i / W, g = ng - 1;
            for (int k = ng - 2; k >= 0; k--) /* the lowest-numbered rectangle that holds the unit wins */
                if (x >= e->sg_tl[k] % W && x <= e->sg_br[k] % W && y >= e->sg_tl[k] / W && y <= e->sg_br[k] / W) g = k;
            mu[i] = (uint8_t)g;
        }
        break;
    case 3: { /* box-out: a spiral from the centre, clockwise or counter-clockwise, claims the first in0 units for group 0 */
        memset(mu, 1, (size_t)units);
        int d = e->sg_dir, x = (W - d) / 2, y = (Hm - d) / 2, l = x, r = x, t = y, b = y, dx = d - 1, dy = d;
        for (int k = 0; k < in0;) {
            if (mu[y * W + x]) mu[y * W + x] = 0, k++;
            if (dx == -1 && x == l) {
                l = l > 0 ? l - 1 : 0, x = l, dx = 0, dy = 2 * d - 1;
            } else if (dx == 1 && x == r) {
                r = r < W - 1 ? r + 1 : W - 1, x = r, dx = 0, dy = 1 - 2 * d;
            } else if (dy == -1 && y == t) {
                t = t > 0 ? t - 1 : 0, y = t, dx = 1 - 2 * d, dy = 0;
            } else if (dy == 1 && y == b) {
                b = b < Hm - 1 ? b + 1 : Hm - 1, y = b, dx = 2 * d - 1, dy = 0;
            } else
                x += dx, y += dy;
        }
        break;
    }
    case 4:
        for (int i = 0; i < units; i++) mu[i] = (uint8_t)(i < upper_left ? e->sg_dir : 1 - e->sg_dir);
        break;
    case 5:
        for (int x = 0, k = 0; x < W; x++)
            for (int y = 0; y < Hm; y++, k++) mu[y * W + x] = (uint8_t)(k < upper_left ? e->sg_dir : 1 - e->sg_dir);
        break;
    default: memcpy(mu, e->sg_ids, (size_t)units); break;
    }
    for (int i = 0; i < e->wmb * e->hmb; i++) /* frame pictures of an interlace SPS: a map unit is two macroblock rows high; field pictures: one macroblock (8.2.2.8) */
        e->sgmap[i] = p->interlace_sps && !e->field ? mu[(i / (2 * W)) * W + i % W] : mu[i];
    free(mu);
}
static int sg_next_mb(const enc *e, int addr) {
    if (!e->sgmap) return addr + 1;
    int i = addr + 1, n = e->wmb * e->hmb;
    while (i < n && e->sgmap[i] != e->sgmap[addr]) i++;
    return i;
}

static size_t write_pps(enc *e, uint8_t *dst, size_t cap) {
    const sg_params *p = &e->p;
    size_t bcap = 64 + (p->slice_groups > 1 ? (size_t)sg_map_units(e) : 0);
    uint8_t *buf = (uint8_t *)malloc(bcap);
    sg_bw w;
    sg_bw_init(&w, buf, bcap);
    sg_put_ue(&w, 0);
    sg_put_ue(&w, 0);
    sg_put(&w, (uint32_t)p->cabac, 1);
    sg_put(&w, p->poc_bottom_delta != 0 && p->poc_type != 2, 1); /* bottom_field_pic_order_in_frame_present_flag */
    sg_put_ue(&w, p->slice_groups > 1 ? (uint32_t)(p->slice_groups - 1) : 0); /* num_slice_groups_minus1 */
    if (p->slice_groups > 1) {
        int ng = p->slice_groups;
        sg_put_ue(&w, (uint32_t)p->fmo_type);
        if (p->fmo_type == 0)
            for (int g = 0; g < ng; g++) sg_put_ue(&w, (uint32_t)e->sg_rl[g]);
        else if (p->fmo_type == 2)
            for (int g = 0; g < ng - 1; g++) sg_put_ue(&w, (uint32_t)e->sg_tl[g]), sg_put_ue(&w, (uint32_t)e->sg_br[g]);
        else if (p->fmo_type >= 3 && p->fmo_type <= 5)
            sg_put(&w, (uint32_t)e->sg_dir, 1), sg_put_ue(&w, (uint32_t)(e->sg_rate - 1));
        else if (p->fmo_type == 6) {
            int bits = 0, units = sg_map_units(e);
            while ((1 << bits) < ng) bits++;
            sg_put_ue(&w, (uint32_t)(units - 1));
            for (int i = 0; i < units; i++) sg_put(&w, e->sg_ids[i], bits);
        }
    }
    sg_put_ue(&w, (uint32_t)(p->num_ref_frames - 1));
    sg_put_ue(&w, 0);
    sg_put(&w, p->weighted_pred != 0, 1);
    sg_put(&w, (uint32_t)p->weighted_bipred, 2); /* weighted_bipred_idc */
    sg_put_se(&w, p->qp - 26);
    sg_put_se(&w, 0);
    sg_put_se(&w, p->chroma_qp_offset);
    sg_put(&w, 1, 1); /* deblocking_filter_control_present */
    sg_put(&w, (uint32_t)p->constrained_intra, 1);
    sg_put(&w, 0, 1);
    if (p->profile_idc == 100) {
        sg_put(&w, (uint32_t)p->transform8x8, 1);
        sg_put(&w, 0, 1); /* pic_scaling_matrix_present */
        sg_put_se(&w, p->chroma_qp_offset + (p->transform8x8 && p->chroma_qp_offset < 12 ? 1 : 0));
    }
    sg_trailing(&w);
    size_t n = sg_write_nal(dst, cap, 1, 3, 8, buf, sg_bw_bytes(&w));
    free(buf);
    return n;
}

static void write_slice_header(enc *e, int first_mb, int idr, int frame_num, int idr_id, int poc_lsb) {
    sg_bw *w = &e->bw;
    const sg_params *p = &e->p;
    int is_b = e->slice_type == 1, is_p = e->slice_type == 0 || is_b; /* is_p: a slice with reference lists */
    sg_put_ue(w, (uint32_t)first_mb);
    sg_put_ue(w, is_b ? 6 : (is_p ? 5 : 7)); /* slice_type: all slices of the picture alike */
    sg_put_ue(w, 0);
    sg_put(w, (uint32_t)frame_num, 8);
    if (p->interlace_sps) sg_put(w, (uint32_t)e->field, 1); /* field_pic_flag */
    if (e->field) sg_put(w, (uint32_t)e->bottom, 1);    /* bottom_field_flag */
    if (idr) sg_put_ue(w, (uint32_t)idr_id);
    const int bottom_delta = e->field || p->poc_type == 2 ? 0 : (idr && p->poc_bottom_delta < 0 ? 0 : p->poc_bottom_delta); /* (sent in frame pictures only) */
    if (p->poc_type == 0) sg_put(w, (uint32_t)poc_lsb, 8);
    if (p->poc_type == 0 && p->poc_bottom_delta != 0 && !e->field) sg_put_se(w, bottom_delta); /* delta_pic_order_cnt_bottom */
    if (p->poc_type == 1) sg_put_se(w, e->delta_poc0); /* delta_pic_order_cnt[0] (delta_pic_order_always_zero_flag = 0) */
    if (p->poc_type == 1 && p->poc_bottom_delta != 0 && !e->field) sg_put_se(w, bottom_delta - 1); /* delta_pic_order_cnt[1]: on top of offset_for_top_to_bottom_field = 1 */
    if (is_b) sg_put(w, e->p.direct_temporal ? 0 : 1, 1); /* direct_spatial_mv_pred_flag */
    if (is_p) {
        int over = e->nref_active != p->num_ref_frames || (is_b && e->nref1_active != 1);
        sg_put(w, (uint32_t)over, 1);
        if (over) {
            sg_put_ue(w, (uint32_t)(e->nref_active - 1));
            if (is_b) sg_put_ue(w, (uint32_t)(e->nref1_active - 1));
        }
        sg_put(w, e->n_rplm > 0, 1); /* ref_pic_list_modification_flag_l0 */
        if (e->n_rplm > 0) {
            for (int i = 0; i < e->n_rplm; i++) {
                sg_put_ue(w, (uint32_t)e->rplm[i].idc);
                sg_put_ue(w, (uint32_t)e->rplm[i].val);
            }
            sg_put_ue(w, 3);
        }
        if (is_b) sg_put(w, 0, 1); /* ref_pic_list_modification_flag_l1 */
        if ((p->weighted_pred && !is_b) || (is_b && p->weighted_bipred == 1)) {
            sg_put_ue(w, (uint32_t)e->wp_ld);
            if (!p->mono) sg_put_ue(w, (uint32_t)e->wp_cd); /* chroma_log2_weight_denom and the chroma weights: ChromaArrayType != 0 only */
            for (int i = 0; i < e->nref_active; i++) {
                int lf = e->wp_w[i] != (1 << e->wp_ld) || e->wp_o[i];
                sg_put(w, (uint32_t)lf, 1);
                if (lf) sg_put_se(w, e->wp_w[i]), sg_put_se(w, e->wp_o[i]);
                int cf = e->wp_cw[i][0] != (1 << e->wp_cd) || e->wp_co[i][0] || e->wp_cw[i][1] != (1 << e->wp_cd) || e->wp_co[i][1];
                if (!p->mono) sg_put(w, (uint32_t)cf, 1);
                if (cf)
                    for (int c = 0; c < 2; c++) sg_put_se(w, e->wp_cw[i][c]), sg_put_se(w, e->wp_co[i][c]);
            }
            for (int i = 0; is_b && i < e->nref1_active; i++) { /* list 1 */
                int lf = e->wb_w1[i] != (1 << e->wp_ld) || e->wb_o1[i];
                sg_put(w, (uint32_t)lf, 1);
                if (lf) sg_put_se(w, e->wb_w1[i]), sg_put_se(w, e->wb_o1[i]);
                int cf = e->wb_cw1[i][0] != (1 << e->wp_cd) || e->wb_co1[i][0] || e->wb_cw1[i][1] != (1 << e->wp_cd) || e->wb_co1[i][1];
                if (!p->mono) sg_put(w, (uint32_t)cf, 1);
                if (cf)
                    for (int c = 0; c < 2; c++) sg_put_se(w, e->wb_cw1[i][c]), sg_put_se(w, e->wb_co1[i][c]);
            }
        }
    }
    /* dec_ref_pic_marking() 7.3.3.3 */
    if (e->nal_ref_idc) {
        if (idr) {
            sg_put(w, 0, 1);                    /* no_output_of_prior_pics_flag */
            sg_put(w, (uint32_t)e->idr_lt, 1);  /* long_term_reference_flag */
        } else {
            sg_put(w, e->n_mmco > 0, 1); /* adaptive_ref_pic_marking_mode_flag (0: sliding window) */
            if (e->n_mmco > 0) {
                for (int i = 0; i < e->n_mmco; i++) {
                    int op = e->mmco[i].op;
                    sg_put_ue(w, (uint32_t)op);
                    if (op == 1 || op == 3) sg_put_ue(w, (uint32_t)e->mmco[i].a1); /* difference_of_pic_nums_minus1 */
                    if (op == 2) sg_put_ue(w, (uint32_t)e->mmco[i].a1);            /* long_term_pic_num */
                    if (op == 3 || op == 6) sg_put_ue(w, (uint32_t)e->mmco[i].a2); /* long_term_frame_idx */
                    if (op == 4) sg_put_ue(w, (uint32_t)e->mmco[i].a1);            /* max_long_term_frame_idx_plus1 */
                }
                sg_put_ue(w, 0);
            }
        }
    }
    if (p->cabac && is_p) sg_put_ue(w, (uint32_t)e->init_idc);
    sg_put_se(w, e->slice_qp - p->qp); /* slice_qp_delta (pic_init_qp = p->qp) */
    sg_put_ue(w, (uint32_t)p->deblock_idc);
    if (p->deblock_idc != 1) {
        sg_put_se(w, p->alpha_off_div2);
        sg_put_se(w, p->beta_off_div2);
    }
    if (p->slice_groups > 1 && p->fmo_type >= 3 && p->fmo_type <= 5 && e->sg_cycle_bits) sg_put(w, (uint32_t)e->sg_cycle, e->sg_cycle_bits);
}

/* ------------------------------------------------------------------ picture management (generator side)
 * The generator decides WHAT happens to its pictures (which ones a P picture predicts from and in which order,
 * which ones stop being references or become long-term) and derives the syntax element values from that;
 * a decoder has to get back to the same pictures from the syntax (8.2.4, 8.2.5). */
#define SG_MAX_FN 256 /* log2_max_frame_num_minus4 = 4 */
static uint32_t g_feat;
static int32_t g_pocs[8192];
static int g_npocs;
uint32_t sg_last_features(void) { return g_feat; }
int sg_last_pocs(int32_t *dst, int cap) {
    for (int i = 0; i < g_npocs && i < cap && dst; i++) dst[i] = g_pocs[i];
    return g_npocs;
}
static int picnum(const sg_pic *p, int cur_fn) { return p->frame_num > cur_fn ? p->frame_num - SG_MAX_FN : p->frame_num; }

/* RefPicList0 of the coming P picture into e->refs[] / e->nref_active; with p->rplm a random re-ordering. */
static void plan_ref_list(enc *e) {
    sg_pic *st[6], *lt[6], *init[12];
    int nst = 0, nlt = 0, n = 0, cur_fn = e->cur_frame_num;
    for (int i = 0; i < 6; i++) {
        sg_pic *q = &e->pics[i];
        if (q == e->cur) continue;
        if (e->p.field_pics && (e->fpics[i][0].dropped || e->fpics[i][1].dropped)) continue; /* only frames with both fields marked (8.2.4.2.1) */
        if (q->is_ref == 1) st[nst++] = q;
        if (q->is_ref == 2) lt[nlt++] = q;
    }
    for (int i = 0; i < nst; i++) /* short-term: most recent (largest PicNum) first */
        for (int j = i + 1; j < nst; j++)
            if (picnum(st[j], cur_fn) > picnum(st[i], cur_fn)) {
                sg_pic *t = st[i];
                st[i] = st[j], st[j] = t;
            }
    for (int i = 0; i < nlt; i++) /* long-term: ascending LongTermPicNum */
        for (int j = i + 1; j < nlt; j++)
            if (lt[j]->long_idx < lt[i]->long_idx) {
                sg_pic *t = lt[i];
                lt[i] = lt[j], lt[j] = t;
            }
    for (int i = 0; i < nst; i++) init[n++] = st[i];
    for (int i = 0; i < nlt; i++) init[n++] = lt[i];
    e->nrefs = n;
    e->nref_active = n < e->p.num_ref_frames ? n : e->p.num_ref_frames;
    /* temporal direct maps the co-located block's reference picture into RefPicList0 of the B picture: an anchor must not
     * predict from the one picture that its own arrival pushes out of the sliding window */
    if (e->p.bframes > 0 && e->p.direct_temporal && e->nref_active > 1 && e->nref_active == e->p.num_ref_frames) e->nref_active--;
    /* ... and with reference B pictures two pictures leave the window before the last B picture of the next group is decoded */
    if (e->p.b_pyramid && e->p.direct_temporal && e->nref_active > 2) e->nref_active = 2;
    e->n_rplm = 0;
    sg_pic *final[12];
    int nf = 0;
    /* non-existing frames (frame_num gaps) sit in the window but are never predicted from: when one is among the active
     * entries the list is re-ordered so that the real pictures come first, and cut behind them */
    int nreal = 0, force_real = 0;
    sg_pic *real[12];
    for (int i = 0; i < n; i++) {
        if (!init[i]->nonexist) real[nreal++] = init[i];
        else if (i < e->nref_active) force_real = 1;
    }
    if (force_real) e->nref_active = nreal < e->nref_active ? nreal : e->nref_active;
    if (force_real || (e->p.rplm && !(e->p.bframes > 0 && e->p.direct_temporal) && n >= 2 && rnd(e) % 100 < 75)) { /* (temporal direct: see above) */
        /* the first k entries become k distinct pictures picked from the WHOLE set of reference pictures */
        int k = force_real ? e->nref_active : 1 + (int)(rnd(e) % (uint32_t)(e->nref_active < 3 ? e->nref_active : 3));
        int pred = cur_fn; /* picNumL0Pred */
        for (int c = 0; c < k; c++) {
            sg_pic *t;
            int dup;
            do {
                t = force_real ? real[c] : init[rnd(e) % (uint32_t)n];
                dup = t->nonexist;
                for (int j = 0; j < nf; j++) dup |= final[j] == t;
            } while (dup);
            final[nf++] = t;
            if (t->is_ref == 2) {
                e->rplm[e->n_rplm].idc = 2, e->rplm[e->n_rplm++].val = t->long_idx;
                g_feat |= 1u << 10;
            } else {
                /* go down or up from the predictor, modulo MaxPicNum; both directions reach every picture */
                int fn = t->frame_num, down = (pred - fn + SG_MAX_FN) % SG_MAX_FN, up = (fn - pred + SG_MAX_FN) % SG_MAX_FN;
                int use_up = down == 0 || (up != 0 && rnd(e) % 100 < 30);
                e->rplm[e->n_rplm].idc = use_up ? 1 : 0;
                e->rplm[e->n_rplm++].val = (use_up ? up : down) - 1;
                g_feat |= 1u << (use_up ? 9 : 8);
                pred = fn;
            }
        }
    }
    for (int i = 0; i < n; i++) { /* the rest follows in initial order */
        int used = 0;
        for (int j = 0; j < e->n_rplm; j++) used |= final[j] == init[i];
        if (!used) final[nf++] = init[i];
    }
    for (int i = 0; i < 4; i++) e->refs[i] = i < e->nref_active ? final[i] : NULL;
    for (int i = 0; i < e->nref_active; i++)
        if (e->refs[i]->is_ref == 2) g_feat |= 1u << 11;
}

/* RefPicList0 / RefPicList1 of a B picture (8.2.4.2.3): by PicOrderCnt around the current picture */
static void plan_b_lists(enc *e) {
    sg_pic *past[6], *future[6];
    int np = 0, nf = 0;
    for (int i = 0; i < 6; i++) {
        sg_pic *q = &e->pics[i];
        if (q == e->cur || q->is_ref != 1) continue;
        if (q->poc < e->cur_poc)
            past[np++] = q;
        else
            future[nf++] = q;
    }
    for (int i = 0; i < np; i++) /* nearest past first */
        for (int j = i + 1; j < np; j++)
            if (past[j]->poc > past[i]->poc) {
                sg_pic *t = past[i];
                past[i] = past[j], past[j] = t;
            }
    for (int i = 0; i < nf; i++) /* nearest future first */
        for (int j = i + 1; j < nf; j++)
            if (future[j]->poc < future[i]->poc) {
                sg_pic *t = future[i];
                future[i] = future[j], future[j] = t;
            }
    sg_pic *l0[12], *l1[12];
    int n = 0;
    for (int i = 0; i < np; i++) l0[n++] = past[i];
    for (int i = 0; i < nf; i++) l0[n++] = future[i];
    n = 0;
    for (int i = 0; i < nf; i++) l1[n++] = future[i];
    for (int i = 0; i < np; i++) l1[n++] = past[i];
    if (n > 1) { /* identical lists: the first two entries of list 1 trade places */
        int same = 1;
        for (int i = 0; i < n; i++) same &= l0[i] == l1[i];
        if (same) {
            sg_pic *t = l1[0];
            l1[0] = l1[1], l1[1] = t;
        }
    }
    e->nrefs = n;
    e->nref_active = n < e->p.num_ref_frames ? n : e->p.num_ref_frames;
    e->nref1_active = n < 2 ? n : 1 + (int)(rnd(e) % 2); /* one or two list-1 entries */
    if (e->nref1_active > e->p.num_ref_frames) e->nref1_active = e->p.num_ref_frames;
    if (e->nal_ref_idc && e->p.direct_temporal) { /* a reference B picture (b_pyramid) may become a co-located picture: see plan_ref_list() */
        if (e->nref_active > 2) e->nref_active = 2;
        e->nref1_active = 1;
    }
    for (int i = 0; i < 4; i++) e->refs[i] = i < e->nref_active ? l0[i] : NULL, e->refs1[i] = i < e->nref1_active ? l1[i] : NULL;
    e->n_rplm = 0;
}

/* Marking of the current (reference, non-IDR) picture.  plan_marking() draws a random script of memory management
 * control operations against a copy of the reference state and records the syntax; apply_marking() then performs the
 * same decisions on the pictures themselves once the picture is complete. */
typedef struct {
    int ref[6], lidx[6], cur_ref, cur_lidx, max_lt, clear_all;
} mark_state;
static int ms_count(const mark_state *m, int kind) {
    int n = 0;
    for (int i = 0; i < 6; i++) n += m->ref[i] == kind;
    return n;
}
static int ms_pick(enc *e, const mark_state *m, int kind) { /* a random picture of that kind */
    int k = (int)(rnd(e) % (uint32_t)ms_count(m, kind));
    for (int i = 0; i < 6; i++)
        if (m->ref[i] == kind && k-- == 0) return i;
    return -1;
}
static void ms_free_lidx(mark_state *m, int idx) {
    for (int i = 0; i < 6; i++)
        if (m->ref[i] == 2 && m->lidx[i] == idx) m->ref[i] = 0;
}
static void add_mmco(enc *e, int op, int a1, int a2) {
    e->mmco[e->n_mmco].op = op, e->mmco[e->n_mmco].a1 = a1, e->mmco[e->n_mmco].a2 = a2;
    e->n_mmco++;
    g_feat |= 1u << op;
}
static void plan_marking(enc *e, mark_state *m) {
    const int maxref = e->p.num_ref_frames, cur_fn = e->cur_frame_num;
    memset(m, 0, sizeof(*m));
    for (int i = 0; i < 6; i++) m->ref[i] = &e->pics[i] == e->cur ? 0 : e->pics[i].is_ref, m->lidx[i] = e->pics[i].long_idx;
    m->cur_ref = 1, m->max_lt = e->max_lt;
    e->n_mmco = 0;
    if (!e->p.mmco || maxref < 2 || rnd(e) % 100 < 35) return; /* sliding window */
    if (rnd(e) % 100 < 6) { /* operation 5 stands alone: everything before this picture is dropped */
        add_mmco(e, 5, 0, 0);
        for (int i = 0; i < 6; i++) m->ref[i] = 0;
        m->clear_all = 1, m->max_lt = 0;
        return;
    }
    int nops = 1 + (int)(rnd(e) % 3);
    for (int k = 0; k < nops; k++) {
        int r = (int)(rnd(e) % 100);
        if (m->max_lt == 0) { /* long-term indices must be enabled before they are used */
            add_mmco(e, 4, 2, 0);
            m->max_lt = 2;
        } else if (r < 25 && ms_count(m, 1) > 0) { /* 3: short-term -> long-term */
            int i = ms_pick(e, m, 1), idx = (int)(rnd(e) % (uint32_t)m->max_lt);
            if (m->cur_ref == 2 && idx == m->cur_lidx) { /* never take the index operation 6 has just given to this picture */
                if (m->max_lt < 2) continue;
                idx = (idx + 1) % m->max_lt;
            }
            add_mmco(e, 3, cur_fn - picnum(&e->pics[i], cur_fn) - 1, idx);
            ms_free_lidx(m, idx);
            m->ref[i] = 2, m->lidx[i] = idx;
        } else if (r < 45 && ms_count(m, 1) > 1) { /* 1: short-term -> unused */
            int i = ms_pick(e, m, 1);
            add_mmco(e, 1, cur_fn - picnum(&e->pics[i], cur_fn) - 1, 0);
            m->ref[i] = 0;
        } else if (r < 60 && ms_count(m, 2) > 0) { /* 2: long-term -> unused */
            int i = ms_pick(e, m, 2);
            add_mmco(e, 2, m->lidx[i], 0);
            m->ref[i] = 0;
        } else if (r < 75 && m->cur_ref == 1) { /* 6: the current picture becomes a long-term picture */
            int idx = (int)(rnd(e) % (uint32_t)m->max_lt);
            add_mmco(e, 6, 0, idx);
            ms_free_lidx(m, idx);
            m->cur_ref = 2, m->cur_lidx = idx;
        } else if (r < 85) { /* 4: shrink (or re-state) the range of long-term indices */
            int plus1 = 1 + (int)(rnd(e) % 2);
            if (m->cur_ref == 2 && plus1 <= m->cur_lidx) plus1 = m->cur_lidx + 1; /* the index just given to this picture survives */
            add_mmco(e, 4, plus1, 0);
            for (int i = 0; i < 6; i++)
                if (m->ref[i] == 2 && m->lidx[i] >= plus1) m->ref[i] = 0;
            m->max_lt = plus1;
        }
    }
    /* no sliding window in adaptive mode: the script itself has to make room, and it must leave at least one
     * short-term slot so that a later sliding-window picture finds something to drop */
    while (ms_count(m, 1) + ms_count(m, 2) + 1 > maxref || ms_count(m, 2) + (m->cur_ref == 2) > maxref - 1) {
        if (ms_count(m, 2) + (m->cur_ref == 2) > maxref - 1 && ms_count(m, 2) > 0) {
            int i = ms_pick(e, m, 2);
            add_mmco(e, 2, m->lidx[i], 0);
            m->ref[i] = 0;
        } else if (ms_count(m, 1) > 0) { /* the oldest short-term picture */
            int best = -1;
            for (int i = 0; i < 6; i++)
                if (m->ref[i] == 1 && (best < 0 || picnum(&e->pics[i], cur_fn) < picnum(&e->pics[best], cur_fn))) best = i;
            add_mmco(e, 1, cur_fn - picnum(&e->pics[best], cur_fn) - 1, 0);
            m->ref[best] = 0;
        } else {
            int i = ms_pick(e, m, 2);
            add_mmco(e, 2, m->lidx[i], 0);
            m->ref[i] = 0;
        }
    }
    if (e->n_mmco == 0) add_mmco(e, 4, m->max_lt, 0); /* adaptive mode was chosen: say something harmless */
}
static void apply_marking(enc *e, const mark_state *m, int idr) {
    if (!e->nal_ref_idc) return; /* non-reference picture: nothing changes */
    if (idr) {
        for (int i = 0; i < 6; i++) e->pics[i].is_ref = 0;
        e->cur->is_ref = e->idr_lt ? 2 : 1, e->cur->long_idx = 0;
        e->max_lt = e->idr_lt ? 1 : 0;
        return;
    }
    if (e->n_mmco > 0) {
        for (int i = 0; i < 6; i++)
            if (&e->pics[i] != e->cur) e->pics[i].is_ref = m->ref[i], e->pics[i].long_idx = m->lidx[i];
        e->cur->is_ref = m->cur_ref, e->cur->long_idx = m->cur_lidx;
        e->max_lt = m->max_lt;
        return;
    }
    /* sliding window: with all slots taken the oldest short-term picture goes */
    int n = 0, oldest = -1;
    for (int i = 0; i < 6; i++) {
        sg_pic *q = &e->pics[i];
        if (q == e->cur || !q->is_ref) continue;
        n++;
        if (q->is_ref == 1 && (oldest < 0 || picnum(q, e->cur_frame_num) < picnum(&e->pics[oldest], e->cur_frame_num))) oldest = i;
    }
    if (n >= e->p.num_ref_frames && oldest >= 0) e->pics[oldest].is_ref = 0;
    e->cur->is_ref = 1;
}

/* ------------------------------------------------------------------ field pictures (sg_params::field_pics)
 * RefPicList0 of a P field (8.2.4.2.2 + 8.2.4.2.5): the reference FRAMES in descending FrameNumWrap order -- the frame of
 * the current field included when its first field is a reference --, then fields taken from them alternately, starting with
 * the parity of the current field; a frame that lacks the wanted parity is passed over, and when one parity runs out the
 * rest of the other follows in order. */
static void plan_field_list(enc *e, int second) {
    int fr[6], n = 0, cur_fn = e->cur_frame_num;
    for (int i = 0; i < 6; i++)
        if (e->pics[i].is_ref == 1 && !e->pics[i].nonexist && (&e->pics[i] != e->cur_frame || second)) fr[n++] = i;
    for (int i = 0; i < n; i++) /* most recent first */
        for (int j = i + 1; j < n; j++)
            if (picnum(&e->pics[fr[j]], cur_fn) > picnum(&e->pics[fr[i]], cur_fn)) {
                int t = fr[i];
                fr[i] = fr[j], fr[j] = t;
            }
    sg_pic *same[6], *opp[6], *list[12];
    int ns = 0, no = 0, nl = 0, a = 0, b = 0;
    for (int i = 0; i < n; i++) { /* a frame that lacks a field -- the current frame its second one, others a field marked unused -- is passed over */
        const int whole = &e->pics[fr[i]] != e->cur_frame;
        if (whole && !e->fpics[fr[i]][e->bottom].dropped) same[ns++] = &e->fpics[fr[i]][e->bottom];
        if (!e->fpics[fr[i]][!e->bottom].dropped) opp[no++] = &e->fpics[fr[i]][!e->bottom];
    }
    while (a < ns && b < no) list[nl++] = same[a++], list[nl++] = opp[b++];
    while (a < ns) list[nl++] = same[a++];
    while (b < no) list[nl++] = opp[b++];
    e->nrefs = nl;
    e->nref_active = nl < 4 ? nl : 4;
    if (e->nref_active > 1 && rnd(e) % 4 == 0) e->nref_active = 1 + (int)(rnd(e) % (uint32_t)e->nref_active); /* a shorter list now and then */
    /* temporal direct maps the field a co-located block predicts from into RefPicList0 of the B field (8.4.1.2.3), whose first four
     * entries are the two fields of the nearest earlier frame and of the next one: an anchor's first field keeps to the former, its
     * second field -- whose list holds the first field of its own frame second, which the B field may only find further down -- to entry 0 */
    if (e->p.bframes > 0 && e->p.direct_temporal) e->nref_active = second ? 1 : (e->nref_active < 2 ? e->nref_active : 2);
    e->n_rplm = 0;
    /* list modification (8.2.4.3 on field picture numbers, 8.2.4.1): the first k entries become k distinct fields picked from ALL
     * reference fields -- picNumF = 2 * FrameNumWrap + 1 for a field of this parity, 2 * FrameNumWrap for one of the other;
     * CurrPicNum = 2 * frame_num + 1, MaxPicNum = 2 * MaxFrameNum */
    sg_pic *final[12];
    int nf = 0;
    if (e->p.rplm && e->p.bframes == 0 && nl >= 2 && rnd(e) % 100 < 75) {
        const int k = 1 + (int)(rnd(e) % (uint32_t)(e->nref_active < 3 ? e->nref_active : 3)), max_pic_num = 2 * SG_MAX_FN;
        int pred = 2 * cur_fn + 1;
        for (int c = 0; c < k; c++) {
            sg_pic *t;
            int dup;
            do {
                t = list[rnd(e) % (uint32_t)nl];
                dup = 0;
                for (int j = 0; j < nf; j++) dup |= final[j] == t;
            } while (dup);
            final[nf++] = t;
            const int num = 2 * picnum(&e->pics[t->fid], cur_fn) + (t->parity == e->bottom), num_mod = (num + max_pic_num) % max_pic_num;
            const int down = (pred - num_mod + max_pic_num) % max_pic_num, up = (num_mod - pred + max_pic_num) % max_pic_num;
            const int use_up = down == 0 || (up != 0 && rnd(e) % 100 < 30);
            e->rplm[e->n_rplm].idc = use_up ? 1 : 0;
            e->rplm[e->n_rplm++].val = (use_up ? up : down) - 1;
            g_feat |= 1u << (use_up ? 9 : 8);
            pred = num_mod;
        }
    }
    for (int i = 0; i < nl; i++) { /* the rest follows in initial order */
        int used = 0;
        for (int j = 0; j < e->n_rplm; j++) used |= final[j] == list[i];
        if (!used) final[nf++] = list[i];
    }
    for (int i = 0; i < 4; i++) e->refs[i] = i < e->nref_active ? final[i] : NULL;
}
/* Marking scripts in field pictures (8.2.5.4.1 on field picture numbers): memory_management_control_operation 1 takes single FIELDS
 * out of the reference set -- a frame then lacks a field in every later list --, and since a script replaces the sliding window,
 * the first field of a frame also makes room for its frame when the window is full. */
typedef struct {
    int n, slot[8], par[8];
} field_drops;
static int field_is_ref(const enc *e, int slot, int par, int second) {
    if (e->pics[slot].is_ref != 1 || e->fpics[slot][par].dropped) return 0;
    if (&e->pics[slot] == e->cur_frame) return second && par == !e->bottom; /* of the current frame: its first field, once coded */
    return 1;
}
static void plan_field_marking(enc *e, int second, field_drops *fd) {
    const int cur_fn = e->cur_frame_num, cur_slot = (int)(e->cur_frame - e->pics);
    int gone[6][2];
    memset(gone, 0, sizeof(gone));
    fd->n = 0, e->n_mmco = 0;
    if (!e->p.mmco || rnd(e) % 100 < 45) return; /* sliding window */
#define DROP(s_, p_) (gone[s_][p_] = 1, fd->slot[fd->n] = (s_), fd->par[fd->n++] = (p_), \
                      add_mmco(e, 1, 2 * cur_fn + 1 - (2 * picnum(&e->pics[s_], cur_fn) + ((p_) == e->bottom)) - 1, 0))
    if (rnd(e) % 100 < 65) { /* one field, anywhere */
        int cand[12][2], nc = 0;
        for (int i = 0; i < 6; i++)
            for (int par = 0; par < 2; par++)
                if (field_is_ref(e, i, par, second)) cand[nc][0] = i, cand[nc++][1] = par;
        if (nc > 1) {
            const int k = (int)(rnd(e) % (uint32_t)nc);
            DROP(cand[k][0], cand[k][1]);
        }
    }
    if (!second && e->n_mmco > 0) /* no sliding window now: the oldest frames leave field by field until this frame fits */
        for (;;) {
            int nfr = 0, oldest = -1;
            for (int i = 0; i < 6; i++) {
                if (i == cur_slot) continue;
                const int left = (field_is_ref(e, i, 0, 0) && !gone[i][0]) + (field_is_ref(e, i, 1, 0) && !gone[i][1]);
                if (!left) continue;
                nfr++;
                if (oldest < 0 || picnum(&e->pics[i], cur_fn) < picnum(&e->pics[oldest], cur_fn)) oldest = i;
            }
            if (nfr < e->p.num_ref_frames || oldest < 0 || fd->n > 5) break;
            for (int par = 0; par < 2; par++)
                if (field_is_ref(e, oldest, par, 0) && !gone[oldest][par]) DROP(oldest, par);
        }
#undef DROP
}
static void apply_field_drops(enc *e, const field_drops *fd) {
    for (int k = 0; k < fd->n; k++) e->fpics[fd->slot[k]][fd->par[k]].dropped = 1;
    for (int i = 0; i < 6; i++) /* a frame without a field left is no reference frame any more (the current one is marked right after) */
        if (e->pics[i].is_ref == 1 && e->fpics[i][0].dropped && e->fpics[i][1].dropped) e->pics[i].is_ref = 0;
    e->cur_frame->is_ref = 1;
}

/* RefPicList0 / RefPicList1 of a B field (8.2.4.2.4 + 8.2.4.2.5): the reference frames by PicOrderCnt around the current field
 * (list 0: at or before it, nearest first, then the later ones; list 1 the other way round), each list then turned into fields
 * by the same alternation as for P fields */
static int alternate_fields(enc *e, const int *fr, int n, sg_pic **list) {
    int nl = 0, a = 0, b = 0;
    while (a < n && b < n) list[nl++] = &e->fpics[fr[a++]][e->bottom], list[nl++] = &e->fpics[fr[b++]][!e->bottom];
    return nl;
}
static void plan_b_field_lists(enc *e) {
    int past[6], future[6], np = 0, nf = 0;
    for (int i = 0; i < 6; i++) {
        if (e->pics[i].is_ref != 1 || &e->pics[i] == e->cur_frame) continue;
        if (e->pics[i].poc <= e->cur_poc) past[np++] = i;
        else future[nf++] = i;
    }
    for (int i = 0; i < np; i++)
        for (int j = i + 1; j < np; j++)
            if (e->pics[past[j]].poc > e->pics[past[i]].poc) {
                int t = past[i];
                past[i] = past[j], past[j] = t;
            }
    for (int i = 0; i < nf; i++)
        for (int j = i + 1; j < nf; j++)
            if (e->pics[future[j]].poc < e->pics[future[i]].poc) {
                int t = future[i];
                future[i] = future[j], future[j] = t;
            }
    int f0[12], f1[12], n = 0;
    for (int i = 0; i < np; i++) f0[n++] = past[i];
    for (int i = 0; i < nf; i++) f0[n++] = future[i];
    n = 0;
    for (int i = 0; i < nf; i++) f1[n++] = future[i];
    for (int i = 0; i < np; i++) f1[n++] = past[i];
    sg_pic *l0[24], *l1[24];
    const int n0 = alternate_fields(e, f0, n, l0), n1 = alternate_fields(e, f1, n, l1);
    if (n1 > 1 && n0 == n1) {
        int same = 1;
        for (int i = 0; i < n1; i++) same &= l0[i] == l1[i];
        if (same) {
            sg_pic *t = l1[0];
            l1[0] = l1[1], l1[1] = t;
        }
    }
    e->nrefs = n0;
    e->nref_active = n0 < 4 ? n0 : 4;
    if (e->nref_active > 1 && rnd(e) % 3 == 0 && !e->p.direct_temporal) e->nref_active = 1 + (int)(rnd(e) % (uint32_t)e->nref_active);
    e->nref1_active = n1 < 2 ? n1 : 1 + (int)(rnd(e) % 3u);
    for (int i = 0; i < 4; i++) e->refs[i] = i < e->nref_active ? l0[i] : NULL, e->refs1[i] = i < e->nref1_active ? l1[i] : NULL;
    e->n_rplm = 0;
}

/* the finished frame in its other shape: woven from its two fields, or split into them */
static void weave_or_split(enc *e, int slot, int from_fields) {
    sg_pic *f = &e->pics[slot];
    for (int par = 0; par < 2; par++) {
        sg_pic *q = &e->fpics[slot][par];
        for (int pl = 0; pl < 3; pl++) {
            const int pw = pl ? e->W / 2 : e->W, ph = pl ? e->fH / 2 : e->fH;
            for (int y = par; y < ph; y += 2) {
                uint8_t *fr = f->pl[pl] + (size_t)y * pw, *fl = q->pl[pl] + (size_t)(y >> 1) * pw;
                if (from_fields) memcpy(fr, fl, (size_t)pw);
                else memcpy(fl, fr, (size_t)pw);
            }
        }
        if (!from_fields) q->id = e->next_id++, q->frame_num = f->frame_num, q->poc = f->poc, q->parity = par, q->fid = slot;
    }
    if (from_fields) f->id = e->next_id++;
}

/* ------------------------------------------------------------------ top level */
size_t sg_encode(const sg_params *pp, uint8_t *stream, size_t cap, uint8_t *recon, size_t recon_cap, uint32_t *frame_sizes) {
    enc *e = (enc *)calloc(1, sizeof(enc));
    size_t out = 0;
    e->p = *pp;
    sg_params *p = &e->p;
    g_err[0] = 0;
    if (p->num_ref_frames < 1) p->num_ref_frames = 1;
    if (p->num_ref_frames > 4) p->num_ref_frames = 4;
    if (p->slices < 1) p->slices = 1;
    if (p->fn_gap_period > 0 && p->num_ref_frames < 3) p->num_ref_frames = 3; /* up to two non-existing frames enter the window: a real picture must survive them */
    if (p->profile_idc != 100) p->transform8x8 = 0, p->scaling_matrix = 0;
    if (p->profile_idc == 66) p->cabac = 0, p->weighted_pred = 0, p->bframes = 0;
    if (p->bframes > 0) { /* B pictures: two anchors must be referable, and the output order differs from the coding order */
        if (p->num_ref_frames < 2) p->num_ref_frames = 2;
        p->poc_type = 0, p->nonref_period = 0, p->mmco = 0, p->idr_long_term = 0;
        if (p->bframes > 3) p->bframes = 3;
        if (p->bframes < 2) p->b_pyramid = 0;
        /* the coding-order planner below puts an anchor every (bframes + 1) pictures: an IDR picture must be one of them */
        if (p->idr_period > 0 && p->idr_period % (p->bframes + 1)) p->idr_period += (p->bframes + 1) - p->idr_period % (p->bframes + 1);
        if (p->b_pyramid) p->num_ref_frames = 4; /* two anchors, the reference B picture of this group and the one of the group before */
    } else
        p->b_pyramid = 0;
    /* a bottom field that comes first moves every picture but the IDR pictures (their PicOrderCnt is 0 by rule): by one only, so that the picture
     * sent with top count 2 still follows its IDR picture in output order and no two frames share a PicOrderCnt (8.2.1) */
    if (p->poc_bottom_delta < 0) p->poc_bottom_delta = -1;
    if (p->mono && p->profile_idc != 100) {
        snprintf(g_err, sizeof(g_err), "mono (chroma_format_idc 0) needs High profile");
        free(e);
        return 0;
    }
    if (p->field_pics) { /* PAFF, every frame as two fields: see sg.h for what that excludes */
        if (p->profile_idc == 66) { /* (with cabac = 1 the field-coded blocks use the UNPINNED context values of sg_cabac_mn.c: ctxIdx 277..398, 436..459) */
            snprintf(g_err, sizeof(g_err), "field_pics needs Main or High profile");
            free(e);
            return 0;
        }
        p->interlace_sps = 1, p->b_pyramid = 0, p->idr_long_term = 0;
        if (p->bframes > 0) p->mmco = 0;
        p->fn_gap_period = 0;
        if (p->field_pics == 3) p->bframes = 0; /* B fields: only in streams that are all fields (co-located pictures of the same shape) */
    }
    e->W = (p->width + 15) & ~15, e->H = (p->height + 15) & ~15;
    e->wmb = e->W / 16, e->hmb = e->H / 16;
    if (p->interlace_sps && ((e->hmb & 1) || ((e->H - p->height) & 3))) {
        snprintf(g_err, sizeof(g_err), "interlace_sps needs an even number of macroblock rows and a height whose padding is a multiple of 4");
        free(e);
        return 0;
    }
    e->fH = e->H;
    const size_t frame_bytes = (size_t)e->W * e->fH * 3 / 2;
    if (p->slices > (p->field_pics ? e->hmb / 2 : e->hmb)) p->slices = p->field_pics ? e->hmb / 2 : e->hmb;
    if (p->slice_groups < 2) p->slice_groups = 0;
    if (p->slice_groups > 8) p->slice_groups = 8;
    if (p->slice_groups && (p->fmo_type < 0 || p->fmo_type > 6)) p->fmo_type = 1;
    if (p->slice_groups && p->fmo_type >= 3 && p->fmo_type <= 5) p->slice_groups = 2; /* the evolving maps have two groups */
    if (p->slice_groups && p->slices > 4) p->slices = 4;
    if (p->slice_groups) plan_slice_groups(e);
    e->rng = 0x9E3779B97F4A7C15ull ^ ((uint64_t)p->seed * 0xD1B54A32D192ED03ull);
    if (!e->rng) e->rng = 1;
    size_t fsz = frame_bytes;
    for (int i = 0; i < 6; i++) {
        e->pics[i].pl[0] = (uint8_t *)malloc(fsz);
        e->pics[i].pl[1] = e->pics[i].pl[0] + e->W * e->H;
        e->pics[i].pl[2] = e->pics[i].pl[1] + e->W * e->H / 4;
        e->pics[i].w = e->W, e->pics[i].h = e->H, e->pics[i].fid = i;
        for (int par = 0; par < 2 && p->field_pics; par++) { /* its fields: half the rows, planes back to back */
            sg_pic *q = &e->fpics[i][par];
            q->pl[0] = (uint8_t *)malloc(fsz / 2);
            q->pl[1] = q->pl[0] + e->W * e->H / 2;
            q->pl[2] = q->pl[1] + e->W * e->H / 8;
            q->w = e->W, q->h = e->H / 2, q->parity = par, q->fid = i;
        }
    }
    e->mb = (emb *)calloc((size_t)e->wmb * e->hmb, sizeof(emb));
    e->db = (sg_dbmb *)calloc((size_t)e->wmb * e->hmb, sizeof(sg_dbmb));
    e->src = (uint8_t *)malloc(fsz);
    uint8_t *frame_src = p->field_pics ? (uint8_t *)malloc(frame_bytes) : NULL;
    size_t slice_cap = fsz * 2 + 4096;
    uint8_t *rbsp = (uint8_t *)malloc(slice_cap);
    /* scaling lists: flat, or the spec's Default_* lists when scaling_matrix is set */
    memset(e->s4, 16, sizeof(e->s4));
    memset(e->s8, 16, sizeof(e->s8));
    if (p->scaling_matrix) {
        for (int l = 0; l < 6; l++) memcpy(e->s4[l], l < 3 ? sg_default4x4_intra : sg_default4x4_inter, 16);
        memcpy(e->s8[0], sg_default8x8_intra, 64);
        memcpy(e->s8[1], sg_default8x8_inter, 64);
    }
    build_scale(e);
    int frame_num = 0, idr_id = 0, poc = 0, refs_since_reset = 0, since_idr = 0;
    static const int poc1_offsets[2] = {2, 6}; /* offset_for_ref_frame[] of write_sps() */
    g_feat = 0, g_npocs = 0;
    /* coding order.  Without B pictures it is the display order.  With them every (bframes + 1)-th picture is an anchor
     * (I or P) and the pictures between two anchors are B pictures coded right after the later anchor -- unless that anchor
     * is an IDR picture or does not exist any more: then they are P pictures in display order (a closed group). */
    int *disp = (int *)malloc(sizeof(int) * (size_t)p->frames), *is_b = (int *)calloc((size_t)p->frames, sizeof(int));
    int *b_ref = (int *)calloc((size_t)p->frames, sizeof(int));
    {
        int n = 0, g = p->bframes + 1, a = 0;
        while (a < p->frames) {
            disp[n++] = a; /* an anchor that starts a chain of groups */
            for (;;) {
                int nx = a + g, nidr = p->idr_period > 0 && nx % p->idr_period == 0;
                if (p->bframes > 0 && nx < p->frames && !nidr) {
                    disp[n++] = nx; /* the later anchor first, then the B pictures between the two */
                    const int mid = p->b_pyramid ? a + g / 2 : -1; /* b_pyramid: the middle one first, as a reference picture */
                    if (mid > 0) is_b[n] = 1, b_ref[n] = 1, disp[n++] = mid;
                    for (int d = a + 1; d < nx; d++)
                        if (d != mid) is_b[n] = 1, disp[n++] = d;
                    a = nx;
                } else {
                    for (int d = a + 1; d < nx && d < p->frames; d++) disp[n++] = d; /* closed group: P pictures in display order */
                    a = nx;
                    break;
                }
            }
        }
    }
    int idr_disp = 0;
    for (int t = 0; t < p->frames; t++) {
        size_t au_start = out;
        const int dsp = disp[t], bpic = is_b[t];
        const int idr_frame = dsp == 0 || (p->idr_period > 0 && dsp % p->idr_period == 0);
        if (idr_frame) idr_disp = dsp;
        mark_state ms;
        field_drops fd;
        fd.n = 0;
        /* field_pics 1 / 2: every frame as two fields; 3: frame by frame, a frame picture or two field pictures (PAFF proper) */
        const int as_fields = p->field_pics == 3 ? (int)((p->seed * 7u + (unsigned)t * 5u + (unsigned)(t / 3)) % 3u != 0) : p->field_pics != 0;
        e->cur_frame = NULL;
      for (int fld = 0; fld < (as_fields ? 2 : 1); fld++) { /* the picture(s) of frame t: the frame, or its two fields */
        /* only the first field of an IDR frame is an IDR picture; the second one is a P field (it may predict from the first) or an I field */
        const int idr = idr_frame && fld == 0;
        int intra_pic = idr || (idr_frame && fld == 1 && (p->seed + (unsigned)t) % 3 == 0);
        e->field = as_fields, e->bottom = as_fields ? (p->field_pics == 2) ^ fld : 0;
        e->H = as_fields ? e->fH / 2 : e->fH, e->hmb = e->H / 16; /* the PICTURE: a frame or one field */
        sg_set_field_mode(as_fields);
        if (!e->field)
            sg_source_frame(p, dsp, e->src);
        else { /* the lines of this parity, planes back to back at field size */
            if (fld == 0) sg_source_frame(p, dsp, frame_src);
            const uint8_t *fs = frame_src;
            uint8_t *o = e->src;
            for (int pl = 0; pl < 3; pl++) {
                const int pw = pl ? e->W / 2 : e->W, ph = pl ? e->fH / 2 : e->fH;
                for (int y = e->bottom; y < ph; y += 2, o += pw) memcpy(o, fs + (size_t)y * pw, (size_t)pw);
                fs += (size_t)pw * ph;
            }
        }
        if (p->bframes > 0) poc = 2 * (dsp - idr_disp);
        if (idr) {
            frame_num = 0, poc = 0, e->nrefs = 0, refs_since_reset = 0, since_idr = 0;
            size_t n = write_sps(e, stream + out, cap - out);
            out += n;
            n = write_pps(e, stream + out, cap - out);
            out += n;
            for (int i = 0; i < 6; i++) e->pics[i].is_ref = 0;
        }
        /* frame_num gap (8.2.5.2): the skipped values enter the window as non-existing frames, oldest pictures leave */
        if (p->fn_gap_period > 0 && p->bframes == 0 && !p->mmco && !idr && since_idr > 0 && since_idr % p->fn_gap_period == 0) {
            const int skip = 1 + (int)(rnd(e) % 2u);
            for (int k = 0; k < skip; k++) {
                int n = 0, oldest = -1;
                for (int i = 0; i < 6; i++) {
                    if (!e->pics[i].is_ref) continue;
                    n++;
                    if (e->pics[i].is_ref == 1 && (oldest < 0 || picnum(&e->pics[i], frame_num) < picnum(&e->pics[oldest], frame_num))) oldest = i;
                }
                if (n >= p->num_ref_frames && oldest >= 0) e->pics[oldest].is_ref = 0;
                sg_pic *ne = NULL;
                for (int i = 0; i < 6 && !ne; i++)
                    if (!e->pics[i].is_ref) ne = &e->pics[i];
                ne->is_ref = 1, ne->nonexist = 1, ne->frame_num = frame_num, ne->id = e->next_id++, ne->poc = 0;
                frame_num = (frame_num + 1) & (SG_MAX_FN - 1);
                refs_since_reset++;
                g_feat |= 1u << 15;
            }
        }
        /* current picture buffer */
        if (fld == 0) { /* the frame store of this frame */
            e->cur_frame = NULL;
            for (int i = 0; i < 6 && !e->cur_frame; i++)
                if (!e->pics[i].is_ref) e->cur_frame = &e->pics[i];
            e->cur_frame->nonexist = 0, e->cur_frame->frame_num = frame_num, e->cur_frame->poc = poc;
            e->fpics[e->cur_frame - e->pics][0].dropped = e->fpics[e->cur_frame - e->pics][1].dropped = 0;
        }
        e->cur = e->field ? &e->fpics[e->cur_frame - e->pics][e->bottom] : e->cur_frame;
        e->cur->nonexist = 0;
        e->cur->id = e->next_id++;
        e->cur->frame_num = frame_num;
        e->cur_frame_num = frame_num;
        e->slice_type = intra_pic ? 2 : (bpic ? 1 : 0);
        const int pic_poc = poc + fld; /* the count that is SENT (a frame's top field; field pictures: the second field one later) */
        /* poc_bottom_delta < 0: the bottom field of a frame picture is the earlier one, and PicOrderCnt(frame) = Min(top, bottom) moves with it
         * (not in IDR pictures, which send Max(d, 0) so that their PicOrderCnt stays 0) */
        const int early = !e->field && !idr && p->poc_type != 2 && p->poc_bottom_delta < 0 ? p->poc_bottom_delta : 0;
        e->cur->poc = e->cur_poc = pic_poc + early;
        if (fld == 0) e->cur_frame->poc = e->cur_poc;
        e->nal_ref_idc = bpic ? (b_ref[t] ? 2 : 0) : ((!idr && p->nonref_period > 1 && since_idr % p->nonref_period == p->nonref_period - 1) ? 0 : 3);
        if (!e->nal_ref_idc) g_feat |= 1u << 12;
        e->idr_lt = idr && p->idr_long_term;
        e->n_rplm = e->n_mmco = 0;
        memset(&ms, 0, sizeof(ms));
        if (bpic && e->field)
            plan_b_field_lists(e);
        else if (bpic)
            plan_b_lists(e);
        else if (e->field) {
            if (!intra_pic) plan_field_list(e, fld);
            if (!idr && e->nal_ref_idc) plan_field_marking(e, fld, &fd);
        } else if (!idr) {
            plan_ref_list(e);
            if (e->nal_ref_idc && !p->field_pics) plan_marking(e, &ms); /* (field streams: scripts in the field pictures only) */
        }
        /* field streams with marking scripts: a frame picture may find no frame with both fields still marked (and a field, in principle,
         * no field at all) -- such a picture is coded as a non-IDR I picture */
        if (p->field_pics && !bpic && !intra_pic && e->nref_active == 0) intra_pic = 1, e->slice_type = 2;
        if (getenv("SG_DEBUG")) {
            fprintf(stderr, "t=%d fn=%d ref_idc=%d list:", t, frame_num, e->nal_ref_idc);
            for (int i = 0; i < e->nref_active && !idr; i++) fprintf(stderr, " id%d(fn%d,%s%d)", e->refs[i]->id, e->refs[i]->frame_num, e->refs[i]->is_ref == 2 ? "L" : "s", e->refs[i]->long_idx);
            fprintf(stderr, " rplm:");
            for (int i = 0; i < e->n_rplm; i++) fprintf(stderr, " (%d,%d)", e->rplm[i].idc, e->rplm[i].val);
            fprintf(stderr, " mmco:");
            for (int i = 0; i < e->n_mmco; i++) fprintf(stderr, " (%d,%d,%d)", e->mmco[i].op, e->mmco[i].a1, e->mmco[i].a2);
            fprintf(stderr, "\n");
        }
        /* pic_order_cnt_type 1 (8.2.1.2): the picture shall come out at POC `poc`; what the offsets of the SPS do not
         * give is sent as delta_pic_order_cnt[0] */
        e->delta_poc0 = 0;
        if (p->poc_type == 1) {
            int abs_fn = e->nal_ref_idc ? refs_since_reset : refs_since_reset - 1, expected = 0;
            if (idr) abs_fn = 0;
            for (int i = 0; i < abs_fn; i++) expected += poc1_offsets[i % 2];
            if (!e->nal_ref_idc) expected += -1; /* offset_for_non_ref_pic */
            e->delta_poc0 = pic_poc - expected - (e->field && e->bottom ? 1 : 0); /* bottom fields: + offset_for_top_to_bottom_field */
            if (e->delta_poc0) g_feat |= 1u << 14;
        }
        if (g_npocs < 8192 && fld == 0) /* (per frame: the smaller of its fields' counts)  pic_order_cnt_type 2 leaves no choice: 2 * FrameNum, minus 1 for non-reference pictures (8.2.1.3) */
            g_pocs[g_npocs++] = p->poc_type == 2 ? (idr ? 0 : 2 * refs_since_reset - (e->nal_ref_idc ? 0 : 1)) : e->cur_poc;
        /* weighted_pred 2: both denominators 7, so that the DEFAULT weight of an entry without a flag is 128 -- outside the
         * range of a coded weight (-128..127); coded weights stay below it */
        const int wld = p->weighted_pred == 2 ? 7 : 5, wcd = p->weighted_pred == 2 ? 7 : 4;
        if (!idr && p->weighted_pred) {
            e->wp_ld = wld, e->wp_cd = wcd;
            for (int i = 0; i < 4; i++) {
                e->wp_w[i] = (1 << wld) + (wld == 7 ? rnd_range(e, -9, -1) : rnd_range(e, -3, 3)), e->wp_o[i] = rnd_range(e, -2, 2);
                for (int c = 0; c < 2; c++) e->wp_cw[i][c] = (1 << wcd) + (wcd == 7 ? rnd_range(e, -5, -1) : rnd_range(e, -1, 1)), e->wp_co[i][c] = rnd_range(e, -1, 1);
            }
            e->wp_w[0] = 1 << wld, e->wp_o[0] = 0; /* first entry default: exercises the flag=0 path */
        }
        if (bpic) { /* explicit weights of both lists (weighted_bipred_idc 1); unit weights otherwise, so that b_predict can index them blindly */
            e->wp_ld = wld, e->wp_cd = wcd;
            for (int i = 0; i < 4; i++) {
                int ex = p->weighted_bipred == 1;
                e->wp_w[i] = (1 << wld) + (ex ? (wld == 7 ? rnd_range(e, -9, -1) : rnd_range(e, -3, 3)) : 0), e->wp_o[i] = ex ? rnd_range(e, -2, 2) : 0;
                e->wb_w1[i] = (1 << wld) + (ex ? (wld == 7 ? rnd_range(e, -9, -1) : rnd_range(e, -3, 3)) : 0), e->wb_o1[i] = ex ? rnd_range(e, -2, 2) : 0;
                for (int c = 0; c < 2; c++) {
                    e->wp_cw[i][c] = (1 << wcd) + (ex ? (wcd == 7 ? rnd_range(e, -5, -1) : rnd_range(e, -1, 1)) : 0), e->wp_co[i][c] = ex ? rnd_range(e, -1, 1) : 0;
                    e->wb_cw1[i][c] = (1 << wcd) + (ex ? (wcd == 7 ? rnd_range(e, -5, -1) : rnd_range(e, -1, 1)) : 0), e->wb_co1[i][c] = ex ? rnd_range(e, -1, 1) : 0;
                }
            }
            e->wb_w1[0] = 1 << wld, e->wb_o1[0] = 0;
        }
        if (p->mono) /* no chroma weights in the stream: the planes of 128 stay 128 under the inferred unit weights */
            for (int i = 0; i < 4; i++)
                for (int c = 0; c < 2; c++) e->wp_cw[i][c] = e->wb_cw1[i][c] = 1 << e->wp_cd, e->wp_co[i][c] = e->wb_co1[i][c] = 0;
        for (int i = 0; i < e->wmb * e->hmb; i++) e->mb[i].type = T_NONE;
        /* the slices of this picture: (first macroblock, macroblock count), in the order they will be sent */
        int sl_first[512], sl_count[512], nsl = 0;
        if (p->slice_groups > 1) {
            int maxc = (sg_map_units(e) + e->sg_rate - 1) / e->sg_rate;
            e->sg_cycle = (3 * t + 1) % (maxc + 1); /* types 3..5: the boundary moves (and, at 0 / max, one group is empty) */
            build_slice_group_map(e, e->sg_cycle);
            for (int g = 0; g < p->slice_groups; g++) {
                int n = 0, first = -1;
                for (int i = 0; i < e->wmb * e->hmb; i++)
                    if (e->sgmap[i] == g) {
                        if (first < 0) first = i;
                        n++;
                    }
                for (int k = 0, a = first, done = 0; k < p->slices && n > 0; k++) { /* the group in p->slices runs of its own order */
                    int c = n * (k + 1) / p->slices - done;
                    if (c <= 0) continue;
                    sl_first[nsl] = a, sl_count[nsl++] = c;
                    done += c;
                    for (int i = 0; i < c; i++) a = sg_next_mb(e, a);
                }
            }
        } else
            for (int s = 0; s < p->slices; s++) {
                int row0 = e->hmb * s / p->slices, row1 = e->hmb * (s + 1) / p->slices;
                if (row1 > row0) sl_first[nsl] = row0 * e->wmb, sl_count[nsl++] = (row1 - row0) * e->wmb;
            }
        if (p->aso) /* arbitrary slice order: a rotation plus a swap, different for every picture */
            for (int k = nsl - 1; k > 0; k--) {
                int j = (int)((p->seed + 7u * (unsigned)t + 3u * (unsigned)k) % (unsigned)(k + 1)), tf = sl_first[k], tc = sl_count[k];
                sl_first[k] = sl_first[j], sl_count[k] = sl_count[j], sl_first[j] = tf, sl_count[j] = tc;
            }
        for (int s = 0; s < nsl; s++) {
            int first = sl_first[s];
            e->slice_id = s;
            e->init_idc = p->cabac_init_idc >= 0 ? p->cabac_init_idc : (t + s) % 3;
            e->slice_qp = p->qp + p->slice_qp_delta * ((t + s) % 3 - 1);
            e->slice_qp = e->slice_qp < 0 ? 0 : (e->slice_qp > 51 ? 51 : e->slice_qp);
            if (e->slice_qp != p->qp) g_feat |= 1u << 13;
            sg_bw_init(&e->bw, rbsp, slice_cap);
            write_slice_header(e, first, idr, frame_num, idr_id, pic_poc & 255);
            e->qp = e->slice_qp, e->prev_dqp_nz = 0, e->skip_run = 0;
            if (p->cabac) {
                while (!sg_bw_aligned(&e->bw)) sg_put(&e->bw, 1, 1);
                sg_cabac_init_ctx(&e->bw, intra_pic ? 0 : 1 + e->init_idc, e->slice_qp);
                sg_cabac_start(&e->bw);
            }
            for (int addr = first, left = sl_count[s]; left > 0; left--, addr = sg_next_mb(e, addr)) {
                begin_mb(e, addr);
                if (intra_pic)
                    encode_intra(e, 1);
                else if (bpic) {
                    int r = (int)(rnd(e) % 1000), kind, sk = p->bskip_permille;
                    if (r < sk)
                        kind = T_BSKIP;
                    else if (r < sk + sk / 2)
                        kind = T_BDIRECT;
                    else if (r < sk + sk / 2 + p->intra_in_p_permille)
                        kind = T_I4; /* any intra */
                    else if (r < sk + sk / 2 + p->intra_in_p_permille + p->sub8x8_permille)
                        kind = T_P16x8 + (int)(rnd(e) % 3);
                    else
                        kind = T_P16;
                    if (p->cabac) {
                        emb *a = MBA(e), *b = MBB(e);
                        sg_cabac_bin(&e->bw, 24 + (a && !IS_SKIPPED(a->type)) + (b && !IS_SKIPPED(b->type)), kind == T_BSKIP);
                    } else if (kind == T_BSKIP)
                        e->skip_run++;
                    else {
                        sg_put_ue(&e->bw, (uint32_t)e->skip_run);
                        e->skip_run = 0;
                    }
                    if (kind == T_I4)
                        encode_intra(e, 0);
                    else
                        encode_b(e, kind);
                } else {
                    int r = (int)(rnd(e) % 1000), kind;
                    if (r < p->skip_permille)
                        kind = T_SKIP;
                    else if (r < p->skip_permille + p->intra_in_p_permille)
                        kind = T_I4; /* any intra */
                    else if (r < p->skip_permille + p->intra_in_p_permille + p->sub8x8_permille)
                        kind = T_P16x8 + (int)(rnd(e) % 3);
                    else
                        kind = T_P16;
                    if (p->cabac) {
                        emb *a = MBA(e), *b = MBB(e);
                        sg_cabac_bin(&e->bw, 11 + (a && a->type != T_SKIP) + (b && b->type != T_SKIP), kind == T_SKIP);
                    } else if (kind == T_SKIP)
                        e->skip_run++;
                    else {
                        sg_put_ue(&e->bw, (uint32_t)e->skip_run);
                        e->skip_run = 0;
                    }
                    if (kind == T_SKIP)
                        encode_skip(e);
                    else if (kind == T_I4)
                        encode_intra(e, 0);
                    else
                        encode_inter(e, kind);
                }
                end_mb(e);
                if (p->cabac) sg_cabac_terminate(&e->bw, left == 1);
            }
            if (p->cabac)
                while (!sg_bw_aligned(&e->bw)) sg_put(&e->bw, 0, 1);
            else {
                if (e->skip_run) sg_put_ue(&e->bw, (uint32_t)e->skip_run);
                sg_trailing(&e->bw);
            }
            if (e->bw.overflow) {
                snprintf(g_err, sizeof(g_err), "slice buffer overflow");
                out = 0;
                goto done;
            }
            size_t n = sg_write_nal(stream + out, cap - out, p->long_start_code || s == 0, e->nal_ref_idc, idr ? 5 : 1, rbsp, sg_bw_bytes(&e->bw));
            if (!n) {
                snprintf(g_err, sizeof(g_err), "stream buffer too small");
                out = 0;
                goto done;
            }
            out += n;
        }
        sg_deblock(e->cur, e->db, e->wmb, e->hmb);
        if (e->field) {
            e->cur->is_ref = e->nal_ref_idc ? 1 : 0; /* (short-term or not: what colZeroFlag asks of RefPicList1[0]; the window lives on the frames) */
            if (e->nal_ref_idc && p->bframes > 0) { /* a later B field may take this field as its co-located picture */
                if (!e->cur->motion) e->cur->motion = malloc(sizeof(emb) * (size_t)e->wmb * (size_t)(e->fH / 32));
                memcpy(e->cur->motion, e->mb, sizeof(emb) * (size_t)e->wmb * e->hmb);
            }
            /* marking is the frame's business (8.2.5.3 counts frames): the first field does what a frame picture would do, the second
             * field of a reference frame just joins it */
            if (e->n_mmco > 0 && !idr) /* a script of operation 1 on fields: no sliding window */
                apply_field_drops(e, &fd);
            else if (fld == 0) {
                sg_pic *fieldpic = e->cur;
                e->cur = e->cur_frame;
                apply_marking(e, &ms, idr);
                e->cur = fieldpic;
            }
            if (fld == 0) continue;
            weave_or_split(e, (int)(e->cur_frame - e->pics), 1);
            e->cur = e->cur_frame;
        } else {
            apply_marking(e, &ms, idr);
            if (p->field_pics) weave_or_split(e, (int)(e->cur_frame - e->pics), 0);
        }
        if (recon && (size_t)(t + 1) * fsz <= recon_cap) memcpy(recon + (size_t)t * fsz, e->cur_frame->pl[0], fsz);
        if (e->cur->is_ref && p->bframes > 0) { /* a later B picture may take this one as its co-located picture */
            if (!e->cur->motion) e->cur->motion = malloc(sizeof(emb) * (size_t)e->wmb * e->hmb);
            memcpy(e->cur->motion, e->mb, sizeof(emb) * (size_t)e->wmb * e->hmb);
        }
      } /* fld */
        since_idr++;
        poc += 2;
        if (e->nal_ref_idc) {
            frame_num = (frame_num + 1) & (SG_MAX_FN - 1);
            refs_since_reset++;
        }
        if (ms.clear_all) { /* after operation 5 the picture counts as frame_num 0 / POC 0 (7.4.3, 8.2.1) */
            e->cur->frame_num = 0;
            frame_num = 1, refs_since_reset = 1, poc = 2;
            if (p->poc_bottom_delta < 0 && p->poc_type != 2) poc -= p->poc_bottom_delta; /* the next frame's bottom field (top + d) comes 2 after this picture's 0 */
            g_pocs[g_npocs - 1] = 0;
        }
        if (idr_frame) idr_id = (idr_id + 1) & 0xFFFF;
        if (frame_sizes) frame_sizes[t] = (uint32_t)(out - au_start);
    }
done:
    free(disp);
    free(is_b);
    free(b_ref);
    for (int i = 0; i < 6; i++) free(e->pics[i].pl[0]), free(e->pics[i].motion), free(e->fpics[i][0].pl[0]), free(e->fpics[i][1].pl[0]), free(e->fpics[i][0].motion), free(e->fpics[i][1].motion);
    free(e->mb);
    free(e->db);
    free(e->src);
    free(frame_src);
    free(e->sgmap);
    free(e->sg_ids);
    free(rbsp);
    free(e);
    return out;
}

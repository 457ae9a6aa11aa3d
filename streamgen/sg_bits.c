/* streamgen/sg_bits.c -- bit writer, Exp-Golomb writers (9.1), NAL wrapping with emulation
 * prevention (7.4.1.1), CABAC arithmetic ENCODER (9.3.4, informative in the spec). */
#include <string.h>
#include "sg_int.h"

void sg_bw_init(sg_bw *w, uint8_t *buf, size_t cap) {
    memset(w, 0, sizeof(*w));
    w->buf = buf;
    w->cap = cap;
}
static void flush_byte(sg_bw *w, uint8_t b) {
    if (w->pos < w->cap)
        w->buf[w->pos++] = b;
    else
        w->overflow = 1;
}
void sg_put(sg_bw *w, uint32_t v, int n) {
    while (n > 0) {
        int take = n > 8 ? 8 : n;
        uint32_t bits = (v >> (n - take)) & ((1u << take) - 1);
        w->acc = (w->acc << take) | bits;
        w->nacc += take;
        n -= take;
        while (w->nacc >= 8) {
            flush_byte(w, (uint8_t)(w->acc >> (w->nacc - 8)));
            w->nacc -= 8;
        }
    }
}
void sg_put_ue(sg_bw *w, uint32_t v) {
    uint32_t x = v + 1;
    int len = 0;
    while ((x >> len) > 1) len++;
    sg_put(w, 0, len);
    sg_put(w, x, len + 1);
}
void sg_put_se(sg_bw *w, int32_t v) { sg_put_ue(w, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
void sg_put_te(sg_bw *w, int range, uint32_t v) {
    if (range > 1)
        sg_put_ue(w, v);
    else
        sg_put(w, !v, 1);
}
int sg_bw_aligned(sg_bw *w) { return w->nacc == 0; }
void sg_trailing(sg_bw *w) {
    sg_put(w, 1, 1);
    while (w->nacc) sg_put(w, 0, 1);
}
size_t sg_bw_bytes(sg_bw *w) { return w->pos; }

size_t sg_write_nal(uint8_t *dst, size_t cap, int long_sc, int ref_idc, int type, const uint8_t *rbsp, size_t n) {
    size_t o = 0;
    if (cap < n + n / 2 + 8) return 0;
    if (long_sc) dst[o++] = 0;
    dst[o++] = 0;
    dst[o++] = 0;
    dst[o++] = 1;
    dst[o++] = (uint8_t)((ref_idc << 5) | type);
    int zeros = 0;
    for (size_t i = 0; i < n; i++) {
        if (zeros >= 2 && rbsp[i] <= 3) {
            dst[o++] = 3;
            zeros = 0;
        }
        dst[o++] = rbsp[i];
        zeros = rbsp[i] == 0 ? zeros + 1 : 0;
    }
    /* an RBSP ending in 0x00 needs a final 0x03 (cabac_zero_words only); never produced here */
    return o;
}

/* ---------------- CABAC encoder, 9.3.4 ---------------- */
void sg_cabac_init_ctx(sg_bw *w, int set, int slice_qp) {
    int qp = slice_qp < 0 ? 0 : (slice_qp > 51 ? 51 : slice_qp);
    for (int i = 0; i < SG_NCTX; i++) {
        int pre = ((sg_cabac_mn[set][i][0] * qp) >> 4) + sg_cabac_mn[set][i][1];
        pre = pre < 1 ? 1 : (pre > 126 ? 126 : pre);
        w->ctx[i] = pre <= 63 ? (uint8_t)((63 - pre) << 1) : (uint8_t)(((pre - 64) << 1) | 1);
    }
}
void sg_cabac_start(sg_bw *w) {
    w->low = 0;
    w->range = 510;
    w->first_bit = 1;
    w->outstanding = 0;
}
static void put_bit(sg_bw *w, int b) { /* 9.3.4.4 PutBit */
    if (w->first_bit)
        w->first_bit = 0;
    else
        sg_put(w, (uint32_t)b, 1);
    while (w->outstanding > 0) {
        sg_put(w, (uint32_t)(1 - b), 1);
        w->outstanding--;
    }
}
static void renorm(sg_bw *w) { /* 9.3.4.3 RenormE */
    while (w->range < 256) {
        if (w->low < 256)
            put_bit(w, 0);
        else if (w->low >= 512) {
            w->low -= 512;
            put_bit(w, 1);
        } else {
            w->low -= 256;
            w->outstanding++;
        }
        w->range <<= 1;
        w->low <<= 1;
    }
}
void sg_cabac_bin(sg_bw *w, int ctx, int bin) { /* 9.3.4.2 EncodeDecision */
    int p = w->ctx[ctx] >> 1, mps = w->ctx[ctx] & 1;
    uint32_t rlps = sg_range_lps[p][(w->range >> 6) & 3];
    w->range -= rlps;
    if (bin != mps) {
        w->low += w->range;
        w->range = rlps;
        if (p == 0) mps = 1 - mps;
        p = sg_trans_lps[p];
    } else if (p < 62)
        p++;
    w->ctx[ctx] = (uint8_t)((p << 1) | mps);
    renorm(w);
}
void sg_cabac_bypass(sg_bw *w, int bin) { /* 9.3.4.4 EncodeBypass */
    w->low <<= 1;
    if (bin) w->low += w->range;
    if (w->low >= 1024) {
        put_bit(w, 1);
        w->low -= 1024;
    } else if (w->low < 512)
        put_bit(w, 0);
    else {
        w->low -= 512;
        w->outstanding++;
    }
}
void sg_cabac_terminate(sg_bw *w, int bin) { /* 9.3.4.5 EncodeTerminate + EncodeFlush */
    w->range -= 2;
    if (bin) {
        w->low += w->range;
        w->range = 2;
        renorm(w);
        put_bit(w, (w->low >> 9) & 1);
        sg_put(w, ((w->low >> 7) & 3) | 1, 2);
    } else
        renorm(w);
}

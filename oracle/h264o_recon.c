/*
 * oracle/h264o_recon.c -- macroblock reconstruction: scaling + inverse transforms (8.5),
 * intra prediction (8.3), inter prediction (8.4.2), in-loop deblocking (8.7).
 *
 * TEST INFRASTRUCTURE ONLY (see h264o.h).
 *
 * The reference has none of this ("Macroblock to YCbCr image decoding" is a TODO, README.md:10;
 * h264/slice.go:599-828 never calls residual()); the only normative source is the spec.
 */
#include <stdlib.h>
#include <string.h>
#include "h264o_int.h"

/* ------------------------------------------------------------------ LevelScale (8.5.9) */
void h264o_build_level_scale(h264o_decoder *d) {
    const h264o_pps *p = d->apps;
    for (int l = 0; l < 6; l++)
        for (int q = 0; q < 6; q++)
            for (int k = 0; k < 16; k++) {
                int r = h264o_zigzag4x4[k], x = r & 3, y = r >> 2;
                int v = ((x & 1) == 0 && (y & 1) == 0) ? h264o_norm4x4[q][0] : (((x & 1) && (y & 1)) ? h264o_norm4x4[q][1] : h264o_norm4x4[q][2]);
                d->level_scale4[l][q][r] = p->scaling4x4[l][k] * v;
            }
    for (int l = 0; l < 2; l++)
        for (int q = 0; q < 6; q++)
            for (int k = 0; k < 64; k++) {
                int r = h264o_zigzag8x8[k], x = r & 7, y = r >> 3, v;
                if ((x & 3) == 0 && (y & 3) == 0)
                    v = h264o_norm8x8[q][0];
                else if ((x & 1) && (y & 1))
                    v = h264o_norm8x8[q][1];
                else if ((x & 3) == 2 && (y & 3) == 2)
                    v = h264o_norm8x8[q][2];
                else if (((y & 3) == 0 && (x & 1)) || ((y & 1) && (x & 3) == 0))
                    v = h264o_norm8x8[q][3];
                else if (((y & 3) == 0 && (x & 3) == 2) || ((y & 3) == 2 && (x & 3) == 0))
                    v = h264o_norm8x8[q][4];
                else
                    v = h264o_norm8x8[q][5];
                d->level_scale8[l][q][r] = p->scaling8x8[l][k] * v;
            }
}

/* ------------------------------------------------------------------ inverse transforms */
/* 8.5.12.2: rows first, then columns; in/out raster 4x4. */
static void idct4x4(const int *d, int *r) {
    int t[16];
    for (int i = 0; i < 4; i++) {
        const int *s = d + 4 * i;
        int e0 = s[0] + s[2], e1 = s[0] - s[2], e2 = (s[1] >> 1) - s[3], e3 = s[1] + (s[3] >> 1);
        t[4 * i + 0] = e0 + e3;
        t[4 * i + 1] = e1 + e2;
        t[4 * i + 2] = e1 - e2;
        t[4 * i + 3] = e0 - e3;
    }
    for (int j = 0; j < 4; j++) {
        int g0 = t[j] + t[8 + j], g1 = t[j] - t[8 + j], g2 = (t[4 + j] >> 1) - t[12 + j], g3 = t[4 + j] + (t[12 + j] >> 1);
        r[j] = (g0 + g3 + 32) >> 6;
        r[4 + j] = (g1 + g2 + 32) >> 6;
        r[8 + j] = (g1 - g2 + 32) >> 6;
        r[12 + j] = (g0 - g3 + 32) >> 6;
    }
}
static void idct8_1d(const int *s, int stride, int *o, int ostride) {
    int d0 = s[0], d1 = s[stride], d2 = s[2 * stride], d3 = s[3 * stride], d4 = s[4 * stride], d5 = s[5 * stride], d6 = s[6 * stride],
        d7 = s[7 * stride];
    int e0 = d0 + d4, e1 = -d3 + d5 - d7 - (d7 >> 1), e2 = d0 - d4, e3 = d1 + d7 - d3 - (d3 >> 1);
    int e4 = (d2 >> 1) - d6, e5 = -d1 + d7 + d5 + (d5 >> 1), e6 = d2 + (d6 >> 1), e7 = d3 + d5 + d1 + (d1 >> 1);
    int f0 = e0 + e6, f1 = e1 + (e7 >> 2), f2 = e2 + e4, f3 = e3 + (e5 >> 2);
    int f4 = e2 - e4, f5 = (e3 >> 2) - e5, f6 = e0 - e6, f7 = e7 - (e1 >> 2);
    o[0] = f0 + f7;
    o[ostride] = f2 + f5;
    o[2 * ostride] = f4 + f3;
    o[3 * ostride] = f6 + f1;
    o[4 * ostride] = f6 - f1;
    o[5 * ostride] = f4 - f3;
    o[6 * ostride] = f2 - f5;
    o[7 * ostride] = f0 - f7;
}
/* 8.5.13 */
static void idct8x8(const int *d, int *r) {
    int t[64];
    for (int i = 0; i < 8; i++) idct8_1d(d + 8 * i, 1, t + 8 * i, 1);
    for (int j = 0; j < 8; j++) idct8_1d(t + j, 8, r + j, 8);
    for (int i = 0; i < 64; i++) r[i] = (r[i] + 32) >> 6;
}
void h264o_kat_idct4x4(const int16_t *c, int16_t *out) {
    int d[16], r[16];
    for (int i = 0; i < 16; i++) d[i] = c[i];
    idct4x4(d, r);
    for (int i = 0; i < 16; i++) out[i] = (int16_t)r[i];
}
void h264o_kat_idct8x8(const int16_t *c, int16_t *out) {
    int d[64], r[64];
    for (int i = 0; i < 64; i++) d[i] = c[i];
    idct8x8(d, r);
    for (int i = 0; i < 64; i++) out[i] = (int16_t)r[i];
}

/* 8.5.12.1 scaling of a 4x4 block given in scan order; dc_override: DC already scaled (8.5.10/11) */
static void scale4x4(const h264o_decoder *d, const int16_t *scan, int list, int qp, int have_dc, int dc, int *out) {
    const int *ls = d->level_scale4[list][qp % 6];
    int sh = qp / 6;
    for (int k = 0; k < 16; k++) {
        int r = d->scan4[k], c = scan[k];
        if (qp >= 24)
            out[r] = (c * ls[r]) * (1 << (sh - 4));
        else
            out[r] = (c * ls[r] + (1 << (3 - sh))) >> (4 - sh);
    }
    if (have_dc) out[0] = dc;
}
static void add_block(uint8_t *dst, int stride, const int *res, int n) {
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) dst[y * stride + x] = (uint8_t)h264o_clip1(dst[y * stride + x] + res[y * n + x]);
}

/* ------------------------------------------------------------------ intra prediction 8.3 */
/* availability of a neighbouring MB for intra prediction of the current MB (6.4.x + 8.3.1.2:
 * constrained_intra_pred makes Inter neighbours "not available for Intra prediction") */
static int intra_avail(h264o_decoder *d, int mbx, int mby) {
    if (mbx < 0 || mby < 0 || mbx >= d->wmb || mby >= d->hmb) return 0;
    h264o_mb *m = &d->mb[mby * d->wmb + mbx];
    if (m->type == MBT_NONE || m->slice_id != d->slice_id) return 0;
    if (mby * d->wmb + mbx >= d->c.addr) return 0;
    if (d->apps->constrained_intra_pred_flag && MB_IS_INTER(m->type)) return 0;
    return 1;
}

/* generic directional predictors for N = 4 or 8; top[-1..2N-1], left[-1..N-1] (top[-1]==left[-1]) */
static void pred_dir(int mode, int N, const int *top, const int *left, uint8_t *dst, int stride) {
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
            int v;
            switch (mode) {
            case 3: /* Diagonal_Down_Left */
                if (x == N - 1 && y == N - 1)
                    v = (top[2 * N - 2] + 3 * top[2 * N - 1] + 2) >> 2;
                else
                    v = (top[x + y] + 2 * top[x + y + 1] + top[x + y + 2] + 2) >> 2;
                break;
            case 4: /* Diagonal_Down_Right */
                if (x > y)
                    v = (top[x - y - 2] + 2 * top[x - y - 1] + top[x - y] + 2) >> 2;
                else if (x < y)
                    v = (left[y - x - 2] + 2 * left[y - x - 1] + left[y - x] + 2) >> 2;
                else
                    v = (top[0] + 2 * top[-1] + left[0] + 2) >> 2;
                break;
            case 5: { /* Vertical_Right */
                int z = 2 * x - y;
                if (z >= 0 && !(z & 1))
                    v = (top[x - (y >> 1) - 1] + top[x - (y >> 1)] + 1) >> 1;
                else if (z >= 0)
                    v = (top[x - (y >> 1) - 2] + 2 * top[x - (y >> 1) - 1] + top[x - (y >> 1)] + 2) >> 2;
                else if (z == -1)
                    v = (left[0] + 2 * top[-1] + top[0] + 2) >> 2;
                else
                    v = (left[y - 2 * x - 1] + 2 * left[y - 2 * x - 2] + left[y - 2 * x - 3] + 2) >> 2;
                break;
            }
            case 6: { /* Horizontal_Down */
                int z = 2 * y - x;
                if (z >= 0 && !(z & 1))
                    v = (left[y - (x >> 1) - 1] + left[y - (x >> 1)] + 1) >> 1;
                else if (z >= 0)
                    v = (left[y - (x >> 1) - 2] + 2 * left[y - (x >> 1) - 1] + left[y - (x >> 1)] + 2) >> 2;
                else if (z == -1)
                    v = (left[0] + 2 * top[-1] + top[0] + 2) >> 2;
                else
                    v = (top[x - 2 * y - 1] + 2 * top[x - 2 * y - 2] + top[x - 2 * y - 3] + 2) >> 2;
                break;
            }
            case 7: /* Vertical_Left */
                if (!(y & 1))
                    v = (top[x + (y >> 1)] + top[x + (y >> 1) + 1] + 1) >> 1;
                else
                    v = (top[x + (y >> 1)] + 2 * top[x + (y >> 1) + 1] + top[x + (y >> 1) + 2] + 2) >> 2;
                break;
            default: { /* 8: Horizontal_Up */
                int z = x + 2 * y;
                if (z > 2 * N - 3)
                    v = left[N - 1];
                else if (z == 2 * N - 3)
                    v = (left[N - 2] + 3 * left[N - 1] + 2) >> 2;
                else if (!(z & 1))
                    v = (left[y + (x >> 1)] + left[y + (x >> 1) + 1] + 1) >> 1;
                else
                    v = (left[y + (x >> 1)] + 2 * left[y + (x >> 1) + 1] + left[y + (x >> 1) + 2] + 2) >> 2;
            }
            }
            dst[y * stride + x] = (uint8_t)v;
        }
}
/* V / H / DC for an NxN block; returns -1 if the mode needs samples that are not available */
static int pred_vhdc(int mode, int N, const int *top, const int *left, int has_top, int has_left, uint8_t *dst, int stride) {
    if (mode == 0) {
        if (!has_top) return -1;
        for (int y = 0; y < N; y++)
            for (int x = 0; x < N; x++) dst[y * stride + x] = (uint8_t)top[x];
    } else if (mode == 1) {
        if (!has_left) return -1;
        for (int y = 0; y < N; y++)
            for (int x = 0; x < N; x++) dst[y * stride + x] = (uint8_t)left[y];
    } else {
        int s = 0, v, lg = N == 4 ? 2 : (N == 8 ? 3 : 4);
        if (has_top)
            for (int i = 0; i < N; i++) s += top[i];
        if (has_left)
            for (int i = 0; i < N; i++) s += left[i];
        if (has_top && has_left)
            v = (s + N) >> (lg + 1);
        else if (has_top || has_left)
            v = (s + (N >> 1)) >> lg;
        else
            v = 128;
        for (int y = 0; y < N; y++)
            for (int x = 0; x < N; x++) dst[y * stride + x] = (uint8_t)v;
    }
    return 0;
}

static int intra4x4_pred(h264o_decoder *d, uint8_t *dst, int stride, int bx, int by, int mode) {
    h264o_curmb *c = &d->c;
    int has_left = bx > 0 || intra_avail(d, c->mbx - 1, c->mby);
    int has_top = by > 0 || intra_avail(d, c->mbx, c->mby - 1);
    int has_tl, has_tr;
    if (bx > 0 && by > 0)
        has_tl = 1;
    else if (bx > 0)
        has_tl = intra_avail(d, c->mbx, c->mby - 1);
    else if (by > 0)
        has_tl = intra_avail(d, c->mbx - 1, c->mby);
    else
        has_tl = intra_avail(d, c->mbx - 1, c->mby - 1);
    if (by == 0)
        has_tr = bx < 3 ? intra_avail(d, c->mbx, c->mby - 1) : intra_avail(d, c->mbx + 1, c->mby - 1);
    else if (bx == 3)
        has_tr = 0;
    else /* inside the MB: available iff decoded earlier in z-order: not for odd bx on odd by */
        has_tr = !((bx & 1) && (by & 1));
    int tb[14], lb[6];
    int *top = tb + 1, *left = lb + 1;
    for (int i = 0; i < 8; i++) top[i] = 128;
    for (int i = 0; i < 4; i++) left[i] = 128;
    top[-1] = 128;
    if (has_top)
        for (int i = 0; i < 4; i++) top[i] = dst[-stride + i];
    if (has_top && has_tr)
        for (int i = 4; i < 8; i++) top[i] = dst[-stride + i];
    else if (has_top)
        for (int i = 4; i < 8; i++) top[i] = top[3];
    if (has_left)
        for (int i = 0; i < 4; i++) left[i] = dst[i * stride - 1];
    if (has_tl) top[-1] = dst[-stride - 1];
    left[-1] = top[-1];
    if (mode <= 2) return pred_vhdc(mode, 4, top, left, has_top, has_left, dst, stride);
    if (mode > 8) return -1;
    /* availability requirements (8.3.1.2.x) */
    if ((mode == 3 || mode == 7) && !has_top) return -1;
    if ((mode == 4 || mode == 5 || mode == 6) && !(has_top && has_left && has_tl)) return -1;
    if (mode == 8 && !has_left) return -1;
    pred_dir(mode, 4, top, left, dst, stride);
    return 0;
}

static int intra8x8_pred(h264o_decoder *d, uint8_t *dst, int stride, int b8, int mode) {
    h264o_curmb *c = &d->c;
    int x8 = b8 & 1, y8 = b8 >> 1;
    int has_left = x8 || intra_avail(d, c->mbx - 1, c->mby);
    int has_top = y8 || intra_avail(d, c->mbx, c->mby - 1);
    int has_tl, has_tr;
    switch (b8) {
    case 0: has_tl = intra_avail(d, c->mbx - 1, c->mby - 1); has_tr = intra_avail(d, c->mbx, c->mby - 1); break;
    case 1: has_tl = intra_avail(d, c->mbx, c->mby - 1); has_tr = intra_avail(d, c->mbx + 1, c->mby - 1); break;
    case 2: has_tl = intra_avail(d, c->mbx - 1, c->mby); has_tr = 1; break;
    default: has_tl = 1; has_tr = 0;
    }
    int p_top[17], p_left[9]; /* unfiltered, index+1 */
    int *pt = p_top + 1, *pl = p_left + 1;
    for (int i = -1; i < 16; i++) pt[i] = 128;
    for (int i = -1; i < 8; i++) pl[i] = 128;
    if (has_top) {
        for (int i = 0; i < 8; i++) pt[i] = dst[-stride + i];
        if (has_tr)
            for (int i = 8; i < 16; i++) pt[i] = dst[-stride + i];
        else
            for (int i = 8; i < 16; i++) pt[i] = pt[7];
    }
    if (has_left)
        for (int i = 0; i < 8; i++) pl[i] = dst[i * stride - 1];
    if (has_tl) pt[-1] = pl[-1] = dst[-stride - 1];
    /* 8.3.2.2.1 reference sample filtering */
    int f_top[17], f_left[9];
    int *top = f_top + 1, *left = f_left + 1;
    for (int i = -1; i < 16; i++) top[i] = pt[i];
    for (int i = -1; i < 8; i++) left[i] = pl[i];
    if (has_top) {
        top[0] = has_tl ? (pt[-1] + 2 * pt[0] + pt[1] + 2) >> 2 : (3 * pt[0] + pt[1] + 2) >> 2;
        for (int i = 1; i < 15; i++) top[i] = (pt[i - 1] + 2 * pt[i] + pt[i + 1] + 2) >> 2;
        top[15] = (pt[14] + 3 * pt[15] + 2) >> 2;
    }
    if (has_tl) {
        if (!has_top || !has_left) {
            if (has_top)
                top[-1] = (3 * pt[-1] + pt[0] + 2) >> 2;
            else if (has_left)
                top[-1] = (3 * pt[-1] + pl[0] + 2) >> 2;
            else
                top[-1] = pt[-1];
        } else
            top[-1] = (pt[0] + 2 * pt[-1] + pl[0] + 2) >> 2;
    }
    left[-1] = top[-1];
    if (has_left) {
        left[0] = has_tl ? (pl[-1] + 2 * pl[0] + pl[1] + 2) >> 2 : (3 * pl[0] + pl[1] + 2) >> 2;
        for (int i = 1; i < 7; i++) left[i] = (pl[i - 1] + 2 * pl[i] + pl[i + 1] + 2) >> 2;
        left[7] = (pl[6] + 3 * pl[7] + 2) >> 2;
    }
    if (mode <= 2) return pred_vhdc(mode, 8, top, left, has_top, has_left, dst, stride);
    if (mode > 8) return -1;
    if ((mode == 3 || mode == 7) && !has_top) return -1;
    if ((mode == 4 || mode == 5 || mode == 6) && !(has_top && has_left && has_tl)) return -1;
    if (mode == 8 && !has_left) return -1;
    pred_dir(mode, 8, top, left, dst, stride);
    return 0;
}

/* Intra16x16 (8.3.3) N=16 / chroma (8.3.4) N=8 plane prediction */
static void pred_plane(int N, const int *top, const int *left, uint8_t *dst, int stride) {
    int H = 0, V = 0, half = N / 2;
    for (int i = 0; i < half; i++) {
        H += (i + 1) * (top[half + i] - top[half - 2 - i]);
        V += (i + 1) * (left[half + i] - left[half - 2 - i]);
    }
    int a = 16 * (left[N - 1] + top[N - 1]);
    int b = N == 16 ? (5 * H + 32) >> 6 : (34 * H + 32) >> 6;
    int c = N == 16 ? (5 * V + 32) >> 6 : (34 * V + 32) >> 6;
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) dst[y * stride + x] = (uint8_t)h264o_clip1((a + b * (x - (half - 1)) + c * (y - (half - 1)) + 16) >> 5);
}
static void gather_edges(uint8_t *dst, int stride, int N, int has_top, int has_left, int has_tl, int *top, int *left) {
    for (int i = -1; i < N; i++) top[i] = left[i] = 128;
    if (has_top)
        for (int i = 0; i < N; i++) top[i] = dst[-stride + i];
    if (has_left)
        for (int i = 0; i < N; i++) left[i] = dst[i * stride - 1];
    if (has_tl) top[-1] = left[-1] = dst[-stride - 1];
}
static int intra16x16_pred(h264o_decoder *d, uint8_t *dst, int stride, int mode) {
    h264o_curmb *c = &d->c;
    int has_left = intra_avail(d, c->mbx - 1, c->mby), has_top = intra_avail(d, c->mbx, c->mby - 1);
    int has_tl = intra_avail(d, c->mbx - 1, c->mby - 1);
    int tb[17], lb[17];
    gather_edges(dst, stride, 16, has_top, has_left, has_tl, tb + 1, lb + 1);
    if (mode == 3) {
        if (!(has_top && has_left && has_tl)) return -1;
        pred_plane(16, tb + 1, lb + 1, dst, stride);
        return 0;
    }
    return pred_vhdc(mode, 16, tb + 1, lb + 1, has_top, has_left, dst, stride);
}
static int chroma_pred(h264o_decoder *d, uint8_t *dst, int stride, int mode) {
    h264o_curmb *c = &d->c;
    int has_left = intra_avail(d, c->mbx - 1, c->mby), has_top = intra_avail(d, c->mbx, c->mby - 1);
    int has_tl = intra_avail(d, c->mbx - 1, c->mby - 1);
    int tb[9], lb[9];
    int *top = tb + 1, *left = lb + 1;
    gather_edges(dst, stride, 8, has_top, has_left, has_tl, top, left);
    if (mode == 0) { /* DC, per 4x4 chroma block (8.3.4.1-3) */
        for (int b = 0; b < 4; b++) {
            int xo = (b & 1) * 4, yo = (b >> 1) * 4, st = 0, sl = 0, v;
            for (int i = 0; i < 4; i++) st += top[xo + i], sl += left[yo + i];
            int use_t = has_top, use_l = has_left;
            if (b == 1 && has_top) use_l = 0;      /* (xO>0, yO==0): top preferred */
            else if (b == 2 && has_left) use_t = 0; /* (xO==0, yO>0): left preferred */
            if (use_t && use_l)
                v = (st + sl + 4) >> 3;
            else if (use_t)
                v = (st + 2) >> 2;
            else if (use_l)
                v = (sl + 2) >> 2;
            else
                v = 128;
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) dst[(yo + y) * stride + xo + x] = (uint8_t)v;
        }
        return 0;
    }
    if (mode == 1) return pred_vhdc(1, 8, top, left, has_top, has_left, dst, stride); /* horizontal */
    if (mode == 2) return pred_vhdc(0, 8, top, left, has_top, has_left, dst, stride); /* vertical */
    if (!(has_top && has_left && has_tl)) return -1;
    pred_plane(8, top, left, dst, stride);
    return 0;
}

/* ------------------------------------------------------------------ inter prediction 8.4.2.2 */
static inline int refpix(const uint8_t *p, int stride, int w, int h, int x, int y) {
    x = x < 0 ? 0 : (x >= w ? w - 1 : x);
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    return p[y * stride + x];
}
static inline int tap6(int a, int b, int c, int d_, int e, int f) { return a - 5 * b + 20 * c + 20 * d_ - 5 * e + f; }

/* one luma sample at integer (xi,yi) + quarter-sample (xf,yf), 8.4.2.2.1 */
static int luma_sample(const uint8_t *p, int stride, int w, int h, int xi, int yi, int xf, int yf) {
#define R(dx, dy) refpix(p, stride, w, h, xi + (dx), yi + (dy))
#define B1(dy) tap6(R(-2, dy), R(-1, dy), R(0, dy), R(1, dy), R(2, dy), R(3, dy))       /* horizontal, row dy */
#define H1(dx) tap6(R(dx, -2), R(dx, -1), R(dx, 0), R(dx, 1), R(dx, 2), R(dx, 3))       /* vertical, column dx */
    int G = R(0, 0);
    if (!xf && !yf) return G;
    int b = h264o_clip1((B1(0) + 16) >> 5), hh = h264o_clip1((H1(0) + 16) >> 5);
    if (yf == 0) return xf == 2 ? b : (xf == 1 ? (G + b + 1) >> 1 : (R(1, 0) + b + 1) >> 1);
    if (xf == 0) return yf == 2 ? hh : (yf == 1 ? (G + hh + 1) >> 1 : (R(0, 1) + hh + 1) >> 1);
    int s = h264o_clip1((B1(1) + 16) >> 5), m = h264o_clip1((H1(1) + 16) >> 5);
    if (xf == 2 || yf == 2) {
        int j1 = tap6(B1(-2), B1(-1), B1(0), B1(1), B1(2), B1(3));
        int j = h264o_clip1((j1 + 512) >> 10);
        if (xf == 2 && yf == 2) return j;
        if (xf == 2) return yf == 1 ? (b + j + 1) >> 1 : (s + j + 1) >> 1; /* f, q */
        return xf == 1 ? (hh + j + 1) >> 1 : (m + j + 1) >> 1;             /* i, k */
    }
    /* diagonal quarter positions e, g, p, r */
    if (xf == 1 && yf == 1) return (b + hh + 1) >> 1;
    if (xf == 3 && yf == 1) return (b + m + 1) >> 1;
    if (xf == 1 && yf == 3) return (hh + s + 1) >> 1;
    return (m + s + 1) >> 1;
#undef R
#undef B1
#undef H1
}

/* 8.4.2.3: combine the predictions of one sample.  n = 1: a = the single prediction (list `l`), explicit weights when
 * `explicit_`; n = 2: a from list 0, b from list 1 -- default average, explicit or implicit weights (logWD = 5, offsets 0). */
static int weight1(int a, int logwd, int w, int o) { return logwd >= 1 ? h264o_clip1(((a * w + (1 << (logwd - 1))) >> logwd) + o) : h264o_clip1(a * w + o); }
static int weight2(int a, int b, int logwd, int w0, int w1, int o0, int o1) {
    return h264o_clip1(((a * w0 + b * w1 + (1 << logwd)) >> (logwd + 1)) + ((o0 + o1 + 1) >> 1));
}
/* 8.4.2.3.1 implicit bi-prediction weights from the picture order counts */
static void implicit_weights(const h264o_decoder *d, const h264o_pic *p0, const h264o_pic *p1, int *w0, int *w1) {
    int tb = h264o_clip3(-128, 127, d->cur->poc - p0->poc), td = h264o_clip3(-128, 127, p1->poc - p0->poc);
    *w0 = *w1 = 32;
    if (td == 0 || p0->ref == 2 || p1->ref == 2) return;
    int tx = (16384 + abs(td / 2)) / td;
    int dsf = h264o_clip3(-1024, 1023, (tb * tx + 32) >> 6);
    if ((dsf >> 2) < -64 || (dsf >> 2) > 128) return;
    *w0 = 64 - (dsf >> 2), *w1 = dsf >> 2;
}
static void inter_pred_mb(h264o_decoder *d, h264o_mb *m) {
    h264o_curmb *c = &d->c;
    int W = d->wmb * 16, H = d->hmb * 16;
    const h264o_slice_header *sh = &d->sh;
    const int bslice = sh->slice_type == 1;
    /* weighted prediction mode: 0 default, 1 explicit, 2 implicit (B only) */
    const int wmode = bslice ? d->apps->weighted_bipred_idc : (d->apps->weighted_pred_flag ? 1 : 0);
    for (int blk = 0; blk < 16; blk++) {
        int bx = blk & 3, by = blk >> 2, i8 = (by >> 1) * 2 + (bx >> 1);
        int x0 = c->mbx * 16 + bx * 4, y0 = c->mby * 16 + by * 4;
        int cx0 = x0 >> 1, cy0 = y0 >> 1;
        int py[2][16], pc[2][2][4], used[2], refidx[2];
        h264o_pic *rp[2];
        for (int l = 0; l < 2; l++) {
            refidx[l] = l ? m->ref1[i8] : m->ref[i8];
            used[l] = refidx[l] >= 0;
            rp[l] = NULL;
            if (!used[l]) continue;
            rp[l] = refidx[l] <= 32 ? (l ? d->rpl1[refidx[l]] : d->rpl0[refidx[l]]) : NULL;
            if (!rp[l]) rp[l] = d->cur; /* missing reference: conceal with the current picture (never hit on valid streams) */
            int mvx = l ? m->mv1[blk][0] : m->mv[blk][0], mvy = l ? m->mv1[blk][1] : m->mv[blk][1];
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++)
                    py[l][y * 4 + x] = luma_sample(rp[l]->plane[0], rp[l]->stride[0], W, H, x0 + x + (mvx >> 2), y0 + y + (mvy >> 2), mvx & 3, mvy & 3);
            /* chroma 8.4.2.2.2: 2x2 samples per 4x4 luma block, mv in 1/8 chroma sample units */
            /* field pictures (Table 8-9): a reference field of the other parity lies half a frame row away -- the chroma vector moves by
             * a quarter chroma sample, down when the bottom field predicts from a top field, up the other way round */
            if (d->field_pic && rp[l]->parity >= 0 && rp[l]->parity != d->bottom) mvy += d->bottom ? 2 : -2;
            int xf = mvx & 7, yf = mvy & 7;
            for (int pl = 1; pl < 3; pl++) {
                const uint8_t *rc = rp[l]->plane[pl];
                int rs = rp[l]->stride[pl];
                for (int y = 0; y < 2; y++)
                    for (int x = 0; x < 2; x++) {
                        int xi = cx0 + x + (mvx >> 3), yi = cy0 + y + (mvy >> 3);
                        int A = refpix(rc, rs, W / 2, H / 2, xi, yi), B = refpix(rc, rs, W / 2, H / 2, xi + 1, yi);
                        int C = refpix(rc, rs, W / 2, H / 2, xi, yi + 1), D = refpix(rc, rs, W / 2, H / 2, xi + 1, yi + 1);
                        pc[l][pl - 1][y * 2 + x] = ((8 - xf) * (8 - yf) * A + xf * (8 - yf) * B + (8 - xf) * yf * C + xf * yf * D + 32) >> 6;
                    }
            }
        }
        if (!used[0] && !used[1]) { /* no list at all: only in damaged streams -- grey */
            used[0] = 1, refidx[0] = 0;
            for (int i = 0; i < 16; i++) py[0][i] = 128;
            for (int i = 0; i < 4; i++) pc[0][0][i] = pc[0][1][i] = 128;
        }
        /* weights of this block: luma and the two chroma planes */
        int lw = sh->luma_log2_weight_denom, cw = sh->chroma_log2_weight_denom;
        int iw0 = 32, iw1 = 32;
        if (wmode == 2 && used[0] && used[1]) implicit_weights(d, rp[0], rp[1], &iw0, &iw1);
        for (int comp = 0; comp < 3; comp++) {
            int n = comp == 0 ? 16 : 4;
            uint8_t *dst = comp == 0 ? d->cur->plane[0] + y0 * d->cur->stride[0] + x0 : d->cur->plane[comp] + cy0 * d->cur->stride[comp] + cx0;
            int stride = d->cur->stride[comp], wdt = comp == 0 ? 4 : 2;
            for (int i = 0; i < n; i++) {
                int a = comp == 0 ? py[0][i] : pc[0][comp - 1][i], bb = comp == 0 ? py[1][i] : pc[1][comp - 1][i], v;
                if (used[0] && used[1]) {
                    if (wmode == 1) {
                        int w0 = comp ? sh->chroma_weight_l0[refidx[0]][comp - 1] : sh->luma_weight_l0[refidx[0]];
                        int w1 = comp ? sh->chroma_weight_l1[refidx[1]][comp - 1] : sh->luma_weight_l1[refidx[1]];
                        int o0 = comp ? sh->chroma_offset_l0[refidx[0]][comp - 1] : sh->luma_offset_l0[refidx[0]];
                        int o1 = comp ? sh->chroma_offset_l1[refidx[1]][comp - 1] : sh->luma_offset_l1[refidx[1]];
                        v = weight2(a, bb, comp ? cw : lw, w0, w1, o0, o1);
                    } else if (wmode == 2)
                        v = weight2(a, bb, 5, iw0, iw1, 0, 0);
                    else
                        v = (a + bb + 1) >> 1;
                } else {
                    int l = used[0] ? 0 : 1;
                    v = l ? bb : a;
                    if (wmode == 1) {
                        int w = l ? (comp ? sh->chroma_weight_l1[refidx[1]][comp - 1] : sh->luma_weight_l1[refidx[1]])
                                  : (comp ? sh->chroma_weight_l0[refidx[0]][comp - 1] : sh->luma_weight_l0[refidx[0]]);
                        int o = l ? (comp ? sh->chroma_offset_l1[refidx[1]][comp - 1] : sh->luma_offset_l1[refidx[1]])
                                  : (comp ? sh->chroma_offset_l0[refidx[0]][comp - 1] : sh->luma_offset_l0[refidx[0]]);
                        v = weight1(v, comp ? cw : lw, w, o);
                    }
                }
                dst[(i / wdt) * stride + (i % wdt)] = (uint8_t)v;
            }
        }
    }
}

/* ------------------------------------------------------------------ residual 8.5.1-8.5.5, 8.5.10/11 */
static void recon_chroma_residual(h264o_decoder *d, h264o_mb *m, int intra) {
    h264o_curmb *c = &d->c;
    if (!c->cbp_chroma) return;
    for (int cc = 0; cc < 2; cc++) {
        int qp = m->qpc[cc];
        int list = (intra ? 1 : 4) + cc;
        int ls00 = d->level_scale4[list][qp % 6][0];
        /* 8.5.11.1/2: c = [[c0,c1],[c2,c3]], f = A c A, dcC = ((f * LS(0,0)) << (qP/6)) >> 5 */
        int c0 = c->cdc[cc][0], c1 = c->cdc[cc][1], c2 = c->cdc[cc][2], c3 = c->cdc[cc][3];
        int f[4] = {c0 + c1 + c2 + c3, c0 - c1 + c2 - c3, c0 + c1 - c2 - c3, c0 - c1 - c2 + c3};
        uint8_t *base = d->cur->plane[1 + cc] + (c->mby * 8) * d->cur->stride[1 + cc] + c->mbx * 8;
        for (int b = 0; b < 4; b++) {
            int dc = ((f[b] * ls00) * (1 << (qp / 6))) >> 5;
            int blk[16], res[16];
            static const int16_t zero16[16] = {0};
            scale4x4(d, (c->cbp_chroma & 2) ? c->cac[cc][b] : zero16, list, qp, 1, dc, blk);
            idct4x4(blk, res);
            add_block(base + (b >> 1) * 4 * d->cur->stride[1 + cc] + (b & 1) * 4, d->cur->stride[1 + cc], res, 4);
        }
    }
}

static void recon_luma_residual_blocks(h264o_decoder *d, h264o_mb *m, int intra) {
    /* non-Intra16x16, non-intra-NxN path: all prediction is already in the picture */
    h264o_curmb *c = &d->c;
    uint8_t *base = d->cur->plane[0] + (c->mby * 16) * d->cur->stride[0] + c->mbx * 16;
    int stride = d->cur->stride[0];
    for (int b8 = 0; b8 < 4; b8++) {
        if (!(c->cbp_luma & (1 << b8))) continue;
        if (c->t8x8) {
            int blk[64], res[64];
            const int *ls = d->level_scale8[intra ? 0 : 1][m->qp % 6];
            int sh = m->qp / 6;
            for (int k = 0; k < 64; k++) {
                int r = d->scan8[k], v = c->luma8[b8][k];
                blk[r] = m->qp >= 36 ? (v * ls[r]) * (1 << (sh - 6)) : (v * ls[r] + (1 << (5 - sh))) >> (6 - sh);
            }
            idct8x8(blk, res);
            add_block(base + (b8 >> 1) * 8 * stride + (b8 & 1) * 8, stride, res, 8);
        } else
            for (int b4 = 0; b4 < 4; b4++) {
                int idx = b8 * 4 + b4, r = h264o_blk_raster(idx);
                int blk[16], res[16];
                scale4x4(d, c->luma[idx], intra ? 0 : 3, m->qp, 0, 0, blk);
                idct4x4(blk, res);
                add_block(base + (r >> 2) * 4 * stride + (r & 3) * 4, stride, res, 4);
            }
    }
}

void h264o_recon_mb(h264o_decoder *d, h264o_mb *m) {
    h264o_curmb *c = &d->c;
    h264o_pic *p = d->cur;
    int sy = p->stride[0];
    uint8_t *Y = p->plane[0] + (c->mby * 16) * sy + c->mbx * 16;
    uint8_t *Cb = p->plane[1] + (c->mby * 8) * p->stride[1] + c->mbx * 8;
    uint8_t *Cr = p->plane[2] + (c->mby * 8) * p->stride[2] + c->mbx * 8;
    switch (c->type) {
    case MBT_IPCM:
        for (int y = 0; y < 16; y++) memcpy(Y + y * sy, c->pcm + 16 * y, 16);
        for (int y = 0; y < 8; y++) memcpy(Cb + y * p->stride[1], c->pcm + 256 + 8 * y, 8);
        for (int y = 0; y < 8; y++) memcpy(Cr + y * p->stride[2], c->pcm + 320 + 8 * y, 8);
        return;
    case MBT_I4x4:
        for (int idx = 0; idx < 16; idx++) {
            int r = h264o_blk_raster(idx), bx = r & 3, by = r >> 2;
            uint8_t *dst = Y + by * 4 * sy + bx * 4;
            if (intra4x4_pred(d, dst, sy, bx, by, m->ipm[r]) < 0) h264o_fail(d, "mb %d: Intra4x4 mode %d needs unavailable samples", c->addr, m->ipm[r]);
            if (c->cbp_luma & (1 << (idx >> 2))) {
                int blk[16], res[16];
                scale4x4(d, c->luma[idx], 0, m->qp, 0, 0, blk);
                idct4x4(blk, res);
                add_block(dst, sy, res, 4);
            }
        }
        break;
    case MBT_I8x8:
        for (int b8 = 0; b8 < 4; b8++) {
            uint8_t *dst = Y + (b8 >> 1) * 8 * sy + (b8 & 1) * 8;
            int mode = m->ipm[(b8 >> 1) * 8 + (b8 & 1) * 2];
            if (intra8x8_pred(d, dst, sy, b8, mode) < 0) h264o_fail(d, "mb %d: Intra8x8 mode %d needs unavailable samples", c->addr, mode);
            if (c->cbp_luma & (1 << b8)) {
                int blk[64], res[64];
                const int *ls = d->level_scale8[0][m->qp % 6];
                int sh = m->qp / 6;
                for (int k = 0; k < 64; k++) {
                    int r = d->scan8[k], v = c->luma8[b8][k];
                    blk[r] = m->qp >= 36 ? (v * ls[r]) * (1 << (sh - 6)) : (v * ls[r] + (1 << (5 - sh))) >> (6 - sh);
                }
                idct8x8(blk, res);
                add_block(dst, sy, res, 8);
            }
        }
        break;
    case MBT_I16x16: {
        if (intra16x16_pred(d, Y, sy, c->i16mode) < 0) h264o_fail(d, "mb %d: Intra16x16 mode %d needs unavailable samples", c->addr, c->i16mode);
        /* 8.5.10: DC Hadamard + scaling */
        int cm[16], f[16], t[16];
        for (int k = 0; k < 16; k++) cm[d->scan4[k]] = c->i16dc[k];
        for (int i = 0; i < 4; i++) { /* rows: A * c */
            int a = cm[i * 4 + 0], b = cm[i * 4 + 1], cc = cm[i * 4 + 2], dd = cm[i * 4 + 3];
            t[i * 4 + 0] = a + b + cc + dd;
            t[i * 4 + 1] = a + b - cc - dd;
            t[i * 4 + 2] = a - b - cc + dd;
            t[i * 4 + 3] = a - b + cc - dd;
        }
        for (int j = 0; j < 4; j++) {
            int a = t[j], b = t[4 + j], cc = t[8 + j], dd = t[12 + j];
            f[j] = a + b + cc + dd;
            f[4 + j] = a + b - cc - dd;
            f[8 + j] = a - b - cc + dd;
            f[12 + j] = a - b + cc - dd;
        }
        int qp = m->qp, ls00 = d->level_scale4[0][qp % 6][0];
        for (int idx = 0; idx < 16; idx++) {
            int r = h264o_blk_raster(idx), bx = r & 3, by = r >> 2;
            int dc = qp >= 36 ? (f[r] * ls00) * (1 << (qp / 6 - 6)) : (f[r] * ls00 + (1 << (5 - qp / 6))) >> (6 - qp / 6);
            int blk[16], res[16];
            static const int16_t zero16[16] = {0};
            scale4x4(d, (c->cbp_luma & (1 << (idx >> 2))) ? c->luma[idx] : zero16, 0, qp, 1, dc, blk);
            idct4x4(blk, res);
            add_block(Y + by * 4 * sy + bx * 4, sy, res, 4);
        }
        break;
    }
    default: /* inter */
        inter_pred_mb(d, m);
        recon_luma_residual_blocks(d, m, 0);
        recon_chroma_residual(d, m, 0);
        return;
    }
    /* intra chroma */
    if (chroma_pred(d, Cb, p->stride[1], c->chroma_mode) < 0 || chroma_pred(d, Cr, p->stride[2], c->chroma_mode) < 0)
        h264o_fail(d, "mb %d: chroma mode %d needs unavailable samples", c->addr, c->chroma_mode);
    recon_chroma_residual(d, m, 1);
}

/* ------------------------------------------------------------------ deblocking 8.7 */
/* filter one line of samples across an edge; pix points at q0, xs = step across the edge */
static void filter_line(uint8_t *pix, int xs, int bS, int alpha, int beta, int tc0, int chroma) {
    int p0 = pix[-xs], p1 = pix[-2 * xs], q0 = pix[0], q1 = pix[xs];
    if (!(abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta)) return;
    if (bS < 4) {
        int tc;
        if (chroma)
            tc = tc0 + 1;
        else {
            int p2 = pix[-3 * xs], q2 = pix[2 * xs];
            int ap = abs(p2 - p0), aq = abs(q2 - q0);
            tc = tc0 + (ap < beta) + (aq < beta);
            if (ap < beta) pix[-2 * xs] = (uint8_t)(p1 + h264o_clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1));
            if (aq < beta) pix[xs] = (uint8_t)(q1 + h264o_clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1));
        }
        int delta = h264o_clip3(-tc, tc, (((q0 - p0) * 4) + (p1 - q1) + 4) >> 3);
        pix[-xs] = (uint8_t)h264o_clip1(p0 + delta);
        pix[0] = (uint8_t)h264o_clip1(q0 - delta);
    } else if (chroma) {
        pix[-xs] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    } else {
        int p2 = pix[-3 * xs], q2 = pix[2 * xs];
        int ap = abs(p2 - p0), aq = abs(q2 - q0);
        int small = abs(p0 - q0) < ((alpha >> 2) + 2);
        if (ap < beta && small) {
            int p3 = pix[-4 * xs];
            pix[-xs] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            pix[-2 * xs] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
            pix[-3 * xs] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
        } else
            pix[-xs] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        if (aq < beta && small) {
            int q3 = pix[3 * xs];
            pix[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
            pix[xs] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
            pix[2 * xs] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
        } else
            pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    }
}

/* 8.7.2.1 bS for the 4-sample segment between 4x4 blocks pb (in MB mp) and qb (in MB mq) */
/* (field macroblocks: the vertical limit of 4 quarter FRAME samples is 2 quarter field samples) */
static int g_mvy_limit = 4;
static int mv_far(const int16_t *a, const int16_t *b) { return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= g_mvy_limit; }
static int edge_bs(const h264o_mb *mp, int pb, const h264o_mb *mq, int qb, int mb_edge, int field_horizontal) {
    /* bS 4 needs a macroblock edge that is vertical, or lies between frame macroblocks: the horizontal edges of a field picture get 3 */
    if (MB_IS_INTRA(mp->type) || MB_IS_INTRA(mq->type)) return mb_edge && !field_horizontal ? 4 : 3;
    if (((mp->nzmask >> pb) & 1) || ((mq->nzmask >> qb) & 1)) return 2;
    /* different reference pictures or a different number of motion vectors; which list a picture is referenced through does not matter */
    int p8 = (pb >> 3) * 2 + ((pb & 3) >> 1), q8 = (qb >> 3) * 2 + ((qb & 3) >> 1);
    int pr0 = mp->ref[p8] >= 0 ? mp->refid[p8] : -1, pr1 = mp->ref1[p8] >= 0 ? mp->refid1[p8] : -1;
    int qr0 = mq->ref[q8] >= 0 ? mq->refid[q8] : -1, qr1 = mq->ref1[q8] >= 0 ? mq->refid1[q8] : -1;
    const int16_t *pm0 = mp->mv[pb], *pm1 = mp->mv1[pb], *qm0 = mq->mv[qb], *qm1 = mq->mv1[qb];
    int np = (pr0 >= 0) + (pr1 >= 0), nq = (qr0 >= 0) + (qr1 >= 0);
    if (np != nq) return 1;
    if (np == 1) {
        int rp = pr0 >= 0 ? pr0 : pr1, rq = qr0 >= 0 ? qr0 : qr1;
        if (rp != rq) return 1;
        return mv_far(pr0 >= 0 ? pm0 : pm1, qr0 >= 0 ? qm0 : qm1);
    }
    if (!((pr0 == qr0 && pr1 == qr1) || (pr0 == qr1 && pr1 == qr0))) return 1;
    if (pr0 != pr1) /* two different pictures: compare the vectors that point into the same picture */
        return pr0 == qr0 ? (mv_far(pm0, qm0) || mv_far(pm1, qm1)) : (mv_far(pm0, qm1) || mv_far(pm1, qm0));
    /* both vectors into the same picture: filtered unless one of the two pairings matches */
    return (mv_far(pm0, qm0) || mv_far(pm1, qm1)) && (mv_far(pm0, qm1) || mv_far(pm1, qm0));
}

static void deblock_mb(h264o_decoder *d, int mbx, int mby) {
    h264o_mb *mq = &d->mb[mby * d->wmb + mbx];
    if (mq->dbf_idc == 1) return;
    h264o_pic *p = d->cur;
    /* dir 0: vertical edges (filter across x), dir 1: horizontal edges */
    for (int dir = 0; dir < 2; dir++) {
        int has_nb = dir == 0 ? mbx > 0 : mby > 0;
        h264o_mb *mn = has_nb ? (dir == 0 ? mq - 1 : mq - d->wmb) : NULL;
        if (mn && mq->dbf_idc == 2 && mn->slice_id != mq->slice_id) mn = NULL;
        for (int e = 0; e < 4; e++) {
            if (e == 0 && !mn) continue;
            if ((e & 1) && mq->t8x8) continue; /* transform_size_8x8_flag: only edges 0 and 2 for luma */
            const h264o_mb *mp = e == 0 ? mn : mq;
            int bS[4];
            for (int k = 0; k < 4; k++) {
                int qb = dir == 0 ? k * 4 + e : e * 4 + k;
                int pb = dir == 0 ? k * 4 + (e == 0 ? 3 : e - 1) : (e == 0 ? 3 : e - 1) * 4 + k;
                bS[k] = edge_bs(mp, pb, mq, qb, e == 0, d->field_pic && dir == 1);
            }
            if (!(bS[0] | bS[1] | bS[2] | bS[3])) continue;
            /* luma */
            {
                int qpav = (mp->qp + mq->qp + 1) >> 1;
                int ia = h264o_clip3(0, 51, qpav + mq->alpha_off), ib = h264o_clip3(0, 51, qpav + mq->beta_off);
                int alpha = h264o_alpha[ia], beta = h264o_beta[ib];
                int stride = p->stride[0];
                uint8_t *base = p->plane[0] + mby * 16 * stride + mbx * 16;
                for (int i = 0; i < 16; i++) {
                    int bs = bS[i >> 2];
                    if (!bs) continue;
                    uint8_t *pix = dir == 0 ? base + i * stride + e * 4 : base + e * 4 * stride + i;
                    filter_line(pix, dir == 0 ? 1 : stride, bs, alpha, beta, bs < 4 ? h264o_tc0[ia][bs - 1] : 0, 0);
                }
            }
            /* chroma: edges 0 and 2 of the luma grid map to chroma sample 0 and 4 (4:2:0) */
            if (e & 1) continue;
            for (int cc = 0; cc < 2; cc++) {
                int qpav = (mp->qpc[cc] + mq->qpc[cc] + 1) >> 1;
                int ia = h264o_clip3(0, 51, qpav + mq->alpha_off), ib = h264o_clip3(0, 51, qpav + mq->beta_off);
                int alpha = h264o_alpha[ia], beta = h264o_beta[ib];
                int stride = p->stride[1 + cc];
                uint8_t *base = p->plane[1 + cc] + mby * 8 * stride + mbx * 8;
                for (int i = 0; i < 8; i++) {
                    int bs = bS[i >> 1];
                    if (!bs) continue;
                    uint8_t *pix = dir == 0 ? base + i * stride + e * 2 : base + e * 2 * stride + i;
                    filter_line(pix, dir == 0 ? 1 : stride, bs, alpha, beta, bs < 4 ? h264o_tc0[ia][bs - 1] : 0, 1);
                }
            }
        }
    }
}

/* 8.7: macroblocks in raster order; per MB vertical edges left->right, then horizontal top->bottom.
 * NOTE: in an MB with transform_size_8x8_flag the odd luma edges are skipped above, but the chroma
 * edge at chroma sample 4 (luma edge 2) is still filtered. */
void h264o_deblock_picture(h264o_decoder *d) {
    g_mvy_limit = d->field_pic ? 2 : 4;
    for (int mby = 0; mby < d->hmb; mby++)
        for (int mbx = 0; mbx < d->wmb; mbx++) deblock_mb(d, mbx, mby);
}

/* Table 8-12, field scan: down the first column, then column by column with the first two rows leading */
const uint8_t h264o_fieldscan4x4[16] = {0, 4, 1, 8, 12, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15};
#define P8(r, c) ((r) * 8 + (c))
/* Table 8-13, field scan: idx -> position, written as (row, column) */
const uint8_t h264o_fieldscan8x8[64] = {
    P8(0, 0), P8(1, 0), P8(2, 0), P8(0, 1), P8(1, 1), P8(3, 0), P8(4, 0), P8(2, 1),
    P8(0, 2), P8(3, 1), P8(5, 0), P8(6, 0), P8(7, 0), P8(4, 1), P8(1, 2), P8(0, 3),
    P8(2, 2), P8(5, 1), P8(6, 1), P8(7, 1), P8(3, 2), P8(1, 3), P8(0, 4), P8(2, 3),
    P8(4, 2), P8(5, 2), P8(6, 2), P8(7, 2), P8(3, 3), P8(1, 4), P8(0, 5), P8(2, 4),
    P8(4, 3), P8(5, 3), P8(6, 3), P8(7, 3), P8(3, 4), P8(1, 5), P8(0, 6), P8(2, 5),
    P8(4, 4), P8(5, 4), P8(6, 4), P8(7, 4), P8(3, 5), P8(1, 6), P8(2, 6), P8(4, 5),
    P8(5, 5), P8(6, 5), P8(7, 5), P8(3, 6), P8(0, 7), P8(1, 7), P8(4, 6), P8(5, 6),
    P8(6, 6), P8(7, 6), P8(2, 7), P8(3, 7), P8(4, 7), P8(5, 7), P8(6, 7), P8(7, 7)};
#undef P8

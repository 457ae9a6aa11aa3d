/*
 * streamgen/sg_recon.c -- the generator's own reconstruction path (closed-loop encoder side):
 * intra prediction 8.3, sub-sample interpolation 8.4.2.2, scaling + inverse transforms 8.5,
 * deblocking 8.7, plus least-squares quantisation against the decoder's basis functions.
 *
 * Written independently of oracle/ and of the product so that "decoder output == generator
 * reconstruction" is a cross-check of two implementations, not a tautology.  Style differs on
 * purpose: prediction writes into small caller buffers (never in place), interpolation is
 * separable over a clamped window, transforms are done on column-major temporaries.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "sg_int.h"

static inline int clip255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static inline int iclip(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------ intra prediction */
typedef struct {
    int t[17]; /* t[0] = p[-1,-1], t[1+x] = p[x,-1] */
    int l[17]; /* l[0] = p[-1,-1], l[1+y] = p[-1,y] */
} edges;
#define T(x) e->t[(x) + 1]
#define L(y) e->l[(y) + 1]

static void load_edges(const sg_pic *p, int plane, int x, int y, int n, int ntop, const sg_avail *a, edges *e) {
    int stride = plane ? p->w / 2 : p->w;
    const uint8_t *s = p->pl[plane] + y * stride + x;
    for (int i = 0; i < 17; i++) e->t[i] = e->l[i] = 128;
    if (a->top) {
        for (int i = 0; i < n; i++) T(i) = s[i - stride];
        for (int i = n; i < ntop; i++) T(i) = a->topright ? s[i - stride] : s[n - 1 - stride];
    }
    if (a->left)
        for (int i = 0; i < n; i++) L(i) = s[i * stride - 1];
    if (a->topleft) e->t[0] = e->l[0] = s[-stride - 1];
}

int sg_intra_mode_allowed(int kind, int mode, const sg_avail *a) {
    if (kind == 4 || kind == 8) {
        switch (mode) {
        case 0: return a->top;
        case 1: return a->left;
        case 2: return 1;
        case 3:
        case 7: return a->top;
        case 4:
        case 5:
        case 6: return a->top && a->left && a->topleft;
        case 8: return a->left;
        }
        return 0;
    }
    if (kind == 16) return mode == 0 ? a->top : (mode == 1 ? a->left : (mode == 2 ? 1 : (a->top && a->left && a->topleft)));
    /* chroma: 0 DC, 1 horizontal, 2 vertical, 3 plane */
    return mode == 0 ? 1 : (mode == 1 ? a->left : (mode == 2 ? a->top : (a->top && a->left && a->topleft)));
}

/* directional modes 3..8 straight from the formulas of 8.3.1.2.4-9 / 8.3.2.2.5-10 */
static void directional(const edges *e, int n, int mode, uint8_t *pred) {
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            int v = 0;
            if (mode == 3) {
                v = (x == n - 1 && y == n - 1) ? (T(2 * n - 2) + 3 * T(2 * n - 1) + 2) >> 2 : (T(x + y) + 2 * T(x + y + 1) + T(x + y + 2) + 2) >> 2;
            } else if (mode == 4) {
                if (x > y)
                    v = (T(x - y - 2) + 2 * T(x - y - 1) + T(x - y) + 2) >> 2;
                else if (x < y)
                    v = (L(y - x - 2) + 2 * L(y - x - 1) + L(y - x) + 2) >> 2;
                else
                    v = (T(0) + 2 * T(-1) + L(0) + 2) >> 2;
            } else if (mode == 5) {
                int z = 2 * x - y, k = x - (y >> 1);
                if (z < -1)
                    v = (L(y - 2 * x - 1) + 2 * L(y - 2 * x - 2) + L(y - 2 * x - 3) + 2) >> 2;
                else if (z == -1)
                    v = (L(0) + 2 * T(-1) + T(0) + 2) >> 2;
                else if (z % 2 == 0)
                    v = (T(k - 1) + T(k) + 1) >> 1;
                else
                    v = (T(k - 2) + 2 * T(k - 1) + T(k) + 2) >> 2;
            } else if (mode == 6) {
                int z = 2 * y - x, k = y - (x >> 1);
                if (z < -1)
                    v = (T(x - 2 * y - 1) + 2 * T(x - 2 * y - 2) + T(x - 2 * y - 3) + 2) >> 2;
                else if (z == -1)
                    v = (L(0) + 2 * T(-1) + T(0) + 2) >> 2;
                else if (z % 2 == 0)
                    v = (L(k - 1) + L(k) + 1) >> 1;
                else
                    v = (L(k - 2) + 2 * L(k - 1) + L(k) + 2) >> 2;
            } else if (mode == 7) {
                int k = x + (y >> 1);
                v = (y % 2 == 0) ? (T(k) + T(k + 1) + 1) >> 1 : (T(k) + 2 * T(k + 1) + T(k + 2) + 2) >> 2;
            } else {
                int z = x + 2 * y, k = y + (x >> 1), zl = 2 * n - 3;
                if (z > zl)
                    v = L(n - 1);
                else if (z == zl)
                    v = (L(n - 2) + 3 * L(n - 1) + 2) >> 2;
                else if (z % 2 == 0)
                    v = (L(k) + L(k + 1) + 1) >> 1;
                else
                    v = (L(k) + 2 * L(k + 1) + L(k + 2) + 2) >> 2;
            }
            pred[y * n + x] = (uint8_t)v;
        }
}
static void flat_modes(const edges *e, int n, int mode, const sg_avail *a, uint8_t *pred) {
    if (mode == 0) {
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) pred[y * n + x] = (uint8_t)T(x);
    } else if (mode == 1) {
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) pred[y * n + x] = (uint8_t)L(y);
    } else {
        int st = 0, sl = 0, dc;
        for (int i = 0; i < n; i++) st += T(i), sl += L(i);
        if (a->top && a->left)
            dc = (st + sl + n) / (2 * n);
        else if (a->top)
            dc = (st + n / 2) / n;
        else if (a->left)
            dc = (sl + n / 2) / n;
        else
            dc = 128;
        memset(pred, dc, (size_t)n * n);
    }
}
void sg_pred_i4(const sg_pic *p, int x, int y, int mode, const sg_avail *a, uint8_t *pred) {
    edges ed, *e = &ed;
    load_edges(p, 0, x, y, 4, 8, a, e);
    if (mode < 3)
        flat_modes(e, 4, mode, a, pred);
    else
        directional(e, 4, mode, pred);
}
void sg_pred_i8(const sg_pic *p, int x, int y, int mode, const sg_avail *a, uint8_t *pred) {
    edges raw, f, *e = &raw;
    load_edges(p, 0, x, y, 8, 16, a, e);
    f = raw;
    /* 8.3.2.2.1 reference sample filtering */
    if (a->top) {
        f.t[1] = a->topleft ? (T(-1) + 2 * T(0) + T(1) + 2) >> 2 : (3 * T(0) + T(1) + 2) >> 2;
        for (int i = 1; i <= 14; i++) f.t[i + 1] = (T(i - 1) + 2 * T(i) + T(i + 1) + 2) >> 2;
        f.t[16] = (T(14) + 3 * T(15) + 2) >> 2;
    }
    if (a->left) {
        f.l[1] = a->topleft ? (L(-1) + 2 * L(0) + L(1) + 2) >> 2 : (3 * L(0) + L(1) + 2) >> 2;
        for (int i = 1; i <= 6; i++) f.l[i + 1] = (L(i - 1) + 2 * L(i) + L(i + 1) + 2) >> 2;
        f.l[8] = (L(6) + 3 * L(7) + 2) >> 2;
    }
    if (a->topleft) {
        int v;
        if (a->top && a->left)
            v = (T(0) + 2 * T(-1) + L(0) + 2) >> 2;
        else if (a->top)
            v = (3 * T(-1) + T(0) + 2) >> 2;
        else if (a->left)
            v = (3 * T(-1) + L(0) + 2) >> 2;
        else
            v = T(-1);
        f.t[0] = f.l[0] = v;
    }
    if (mode < 3)
        flat_modes(&f, 8, mode, a, pred);
    else
        directional(&f, 8, mode, pred);
}
static void plane_mode(const edges *e, int n, uint8_t *pred) {
    int hh = 0, vv = 0, m = n / 2;
    for (int k = 1; k <= m; k++) {
        hh += k * (T(m - 1 + k) - T(m - 1 - k));
        vv += k * (L(m - 1 + k) - L(m - 1 - k));
    }
    int a = 16 * (L(n - 1) + T(n - 1));
    int b = n == 16 ? (5 * hh + 32) >> 6 : (34 * hh + 32) >> 6;
    int c = n == 16 ? (5 * vv + 32) >> 6 : (34 * vv + 32) >> 6;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) pred[y * n + x] = (uint8_t)clip255((a + b * (x - m + 1) + c * (y - m + 1) + 16) >> 5);
}
void sg_pred_i16(const sg_pic *p, int x, int y, int mode, const sg_avail *a, uint8_t *pred) {
    edges ed, *e = &ed;
    load_edges(p, 0, x, y, 16, 16, a, e);
    if (mode == 3)
        plane_mode(e, 16, pred);
    else
        flat_modes(e, 16, mode, a, pred);
}
void sg_pred_chroma(const sg_pic *p, int plane, int x, int y, int mode, const sg_avail *a, uint8_t *pred) {
    edges ed, *e = &ed;
    load_edges(p, plane, x, y, 8, 8, a, e);
    if (mode == 3) {
        plane_mode(e, 8, pred);
        return;
    }
    if (mode == 1 || mode == 2) {
        flat_modes(e, 8, mode == 1 ? 1 : 0, a, pred);
        return;
    }
    /* DC per 4x4 quadrant: 8.3.4.1 (top-left, bottom-right), .2 (top-right), .3 (bottom-left) */
    for (int q = 0; q < 4; q++) {
        int qx = (q & 1) * 4, qy = (q >> 1) * 4, st = 0, sl = 0, dc;
        for (int i = 0; i < 4; i++) st += T(qx + i), sl += L(qy + i);
        int tfirst = (q == 1), lfirst = (q == 2);
        if (tfirst && a->top)
            dc = (st + 2) >> 2;
        else if (lfirst && a->left)
            dc = (sl + 2) >> 2;
        else if (a->top && a->left)
            dc = (st + sl + 4) >> 3;
        else if (a->top)
            dc = (st + 2) >> 2;
        else if (a->left)
            dc = (sl + 2) >> 2;
        else
            dc = 128;
        for (int yy = 0; yy < 4; yy++)
            for (int xx = 0; xx < 4; xx++) pred[(qy + yy) * 8 + qx + xx] = (uint8_t)dc;
    }
}

/* ------------------------------------------------------------------ inter prediction (separable) */
void sg_mc_luma(const sg_pic *ref, int x, int y, int w, int h, int mvx, int mvy, uint8_t *dst, int dstride) {
    /* clamped integer window covering [-2, +3] around the block */
    int win[21 + 5][21 + 5];
    int fx = mvx & 3, fy = mvy & 3, ix = x + (mvx >> 2), iy = y + (mvy >> 2);
    int ww = w + 5, wh = h + 5;
    for (int j = 0; j < wh; j++) {
        int sy = iclip(iy + j - 2, 0, ref->h - 1);
        for (int i = 0; i < ww; i++) win[j][i] = ref->pl[0][sy * ref->w + iclip(ix + i - 2, 0, ref->w - 1)];
    }
    /* hb[j][i]: horizontal 6-tap intermediate at rows j (all window rows), columns i in [0,w) */
    int hb[26][22], vb[22][26];
    for (int j = 0; j < wh; j++)
        for (int i = 0; i < w; i++)
            hb[j][i] = win[j][i] - 5 * win[j][i + 1] + 20 * win[j][i + 2] + 20 * win[j][i + 3] - 5 * win[j][i + 4] + win[j][i + 5];
    /* vb[j][i]: vertical 6-tap intermediate at rows j in [0,h), all window columns */
    for (int j = 0; j < h; j++)
        for (int i = 0; i < ww; i++)
            vb[j][i] = win[j][i] - 5 * win[j + 1][i] + 20 * win[j + 2][i] + 20 * win[j + 3][i] - 5 * win[j + 4][i] + win[j + 5][i];
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            int G = win[j + 2][i + 2];
            int b = clip255((hb[j + 2][i] + 16) >> 5);      /* half sample right of G */
            int hv = clip255((vb[j][i + 2] + 16) >> 5);     /* half sample below G */
            int s = clip255((hb[j + 3][i] + 16) >> 5);      /* b of the row below */
            int m = clip255((vb[j][i + 3] + 16) >> 5);      /* h of the column to the right */
            int jj = hb[j][i] - 5 * hb[j + 1][i] + 20 * hb[j + 2][i] + 20 * hb[j + 3][i] - 5 * hb[j + 4][i] + hb[j + 5][i];
            int c = clip255((jj + 512) >> 10);
            int v;
            switch (fy * 4 + fx) {
            case 0: v = G; break;
            case 1: v = (G + b + 1) >> 1; break;
            case 2: v = b; break;
            case 3: v = (win[j + 2][i + 3] + b + 1) >> 1; break;
            case 4: v = (G + hv + 1) >> 1; break;
            case 5: v = (b + hv + 1) >> 1; break;
            case 6: v = (b + c + 1) >> 1; break;
            case 7: v = (b + m + 1) >> 1; break;
            case 8: v = hv; break;
            case 9: v = (hv + c + 1) >> 1; break;
            case 10: v = c; break;
            case 11: v = (c + m + 1) >> 1; break;
            case 12: v = (win[j + 3][i + 2] + hv + 1) >> 1; break;
            case 13: v = (hv + s + 1) >> 1; break;
            case 14: v = (c + s + 1) >> 1; break;
            default: v = (m + s + 1) >> 1; break;
            }
            dst[j * dstride + i] = (uint8_t)v;
        }
}
void sg_mc_chroma(const sg_pic *ref, int plane, int x, int y, int w, int h, int mvx, int mvy, uint8_t *dst, int dstride) {
    int cw = ref->w / 2, ch = ref->h / 2, fx = mvx & 7, fy = mvy & 7, ix = x + (mvx >> 3), iy = y + (mvy >> 3);
    const uint8_t *r = ref->pl[plane];
    for (int j = 0; j < h; j++) {
        int y0 = iclip(iy + j, 0, ch - 1), y1 = iclip(iy + j + 1, 0, ch - 1);
        for (int i = 0; i < w; i++) {
            int x0 = iclip(ix + i, 0, cw - 1), x1 = iclip(ix + i + 1, 0, cw - 1);
            int v = (8 - fx) * (8 - fy) * r[y0 * cw + x0] + fx * (8 - fy) * r[y0 * cw + x1] + (8 - fx) * fy * r[y1 * cw + x0] + fx * fy * r[y1 * cw + x1];
            dst[j * dstride + i] = (uint8_t)((v + 32) >> 6);
        }
    }
}

/* ------------------------------------------------------------------ frame / field mode (sg_set_field_mode) */
/* field scans (Table 8-12 / 8-13 field columns), as raster positions x + 4 * y and x + 8 * y */
static const uint8_t field_scan4x4[16] = {0, 4, 1, 8, 12, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15};
#define P8(x, y) ((x) + 8 * (y))
static const uint8_t field_scan8x8[64] = {
    P8(0, 0), P8(0, 1), P8(0, 2), P8(1, 0), P8(1, 1), P8(0, 3), P8(0, 4), P8(1, 2), P8(2, 0), P8(1, 3), P8(0, 5), P8(0, 6), P8(0, 7), P8(1, 4), P8(2, 1), P8(3, 0),
    P8(2, 2), P8(1, 5), P8(1, 6), P8(1, 7), P8(2, 3), P8(3, 1), P8(4, 0), P8(3, 2), P8(2, 4), P8(2, 5), P8(2, 6), P8(2, 7), P8(3, 3), P8(4, 1), P8(5, 0), P8(4, 2),
    P8(3, 4), P8(3, 5), P8(3, 6), P8(3, 7), P8(4, 3), P8(5, 1), P8(6, 0), P8(5, 2), P8(4, 4), P8(4, 5), P8(4, 6), P8(4, 7), P8(5, 3), P8(6, 1), P8(6, 2), P8(5, 4),
    P8(5, 5), P8(5, 6), P8(5, 7), P8(6, 3), P8(7, 0), P8(7, 1), P8(6, 4), P8(6, 5), P8(6, 6), P8(6, 7), P8(7, 2), P8(7, 3), P8(7, 4), P8(7, 5), P8(7, 6), P8(7, 7)};
#undef P8
static const uint8_t *g_scan4 = sg_zigzag4x4, *g_scan8 = sg_zigzag8x8;
static int g_field;
void sg_set_field_mode(int on) {
    g_field = on != 0;
    g_scan4 = on ? field_scan4x4 : sg_zigzag4x4;
    g_scan8 = on ? field_scan8x8 : sg_zigzag8x8;
}

/* ------------------------------------------------------------------ scaling + inverse transforms */
static void inv4(int *m) { /* m raster 4x4, in place: 8.5.12.2 */
    for (int pass = 0; pass < 2; pass++) {
        int step = pass ? 4 : 1, line = pass ? 1 : 4;
        for (int k = 0; k < 4; k++) {
            int *p = m + k * line;
            int a = p[0], b = p[step], c = p[2 * step], d = p[3 * step];
            int s0 = a + c, s1 = a - c, s2 = (b >> 1) - d, s3 = b + (d >> 1);
            p[0] = s0 + s3;
            p[step] = s1 + s2;
            p[2 * step] = s1 - s2;
            p[3 * step] = s0 - s3;
        }
    }
    for (int i = 0; i < 16; i++) m[i] = (m[i] + 32) >> 6;
}
static void inv8(int *m) { /* 8.5.13 */
    for (int pass = 0; pass < 2; pass++) {
        int step = pass ? 8 : 1, line = pass ? 1 : 8;
        for (int k = 0; k < 8; k++) {
            int *p = m + k * line;
            int d0 = p[0], d1 = p[step], d2 = p[2 * step], d3 = p[3 * step], d4 = p[4 * step], d5 = p[5 * step], d6 = p[6 * step], d7 = p[7 * step];
            int a0 = d0 + d4, a2 = d0 - d4, a4 = (d2 >> 1) - d6, a6 = d2 + (d6 >> 1);
            int a1 = d5 - d3 - d7 - (d7 >> 1), a3 = d1 + d7 - d3 - (d3 >> 1), a5 = d7 - d1 + d5 + (d5 >> 1), a7 = d3 + d5 + d1 + (d1 >> 1);
            int b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
            int b1 = a1 + (a7 >> 2), b3 = a3 + (a5 >> 2), b5 = (a3 >> 2) - a5, b7 = a7 - (a1 >> 2);
            p[0] = b0 + b7;
            p[step] = b2 + b5;
            p[2 * step] = b4 + b3;
            p[3 * step] = b6 + b1;
            p[4 * step] = b6 - b1;
            p[5 * step] = b4 - b3;
            p[6 * step] = b2 - b5;
            p[7 * step] = b0 - b7;
        }
    }
    for (int i = 0; i < 64; i++) m[i] = (m[i] + 32) >> 6;
}
void sg_residual4(const int16_t *lev, const int *ls, int qp, int have_dc, int dc, int *res) {
    int per = qp / 6;
    for (int k = 0; k < 16; k++) {
        int pos = g_scan4[k], v = lev[k] * ls[pos];
        res[pos] = per >= 4 ? v * (1 << (per - 4)) : (v + (1 << (3 - per))) >> (4 - per);
    }
    if (have_dc) res[0] = dc;
    inv4(res);
}
void sg_residual8(const int16_t *lev, const int *ls, int qp, int *res) {
    int per = qp / 6;
    for (int k = 0; k < 64; k++) {
        int pos = g_scan8[k], v = lev[k] * ls[pos];
        res[pos] = per >= 6 ? v * (1 << (per - 6)) : (v + (1 << (5 - per))) >> (6 - per);
    }
    inv8(res);
}
static void hadamard4(const int *in, int *out) {
    int t[16];
    for (int r = 0; r < 4; r++) {
        const int *p = in + 4 * r;
        int s01 = p[0] + p[1], d01 = p[0] - p[1], s23 = p[2] + p[3], d23 = p[2] - p[3];
        t[4 * r + 0] = s01 + s23;
        t[4 * r + 1] = s01 - s23;
        t[4 * r + 2] = d01 - d23;
        t[4 * r + 3] = d01 + d23;
    }
    for (int c = 0; c < 4; c++) {
        int p0 = t[c], p1 = t[4 + c], p2 = t[8 + c], p3 = t[12 + c];
        int s01 = p0 + p1, d01 = p0 - p1, s23 = p2 + p3, d23 = p2 - p3;
        out[c] = s01 + s23;
        out[4 + c] = s01 - s23;
        out[8 + c] = d01 - d23;
        out[12 + c] = d01 + d23;
    }
}
void sg_luma_dc(const int16_t *lev_scan, int ls00, int qp, int *dc) { /* 8.5.10 */
    int c[16], f[16], per = qp / 6;
    for (int k = 0; k < 16; k++) c[g_scan4[k]] = lev_scan[k];
    hadamard4(c, f);
    for (int i = 0; i < 16; i++) dc[i] = per >= 6 ? (f[i] * ls00) * (1 << (per - 6)) : (f[i] * ls00 + (1 << (5 - per))) >> (6 - per);
}
void sg_chroma_dc(const int16_t *l, int ls00, int qpc, int *dc) { /* 8.5.11 */
    int f[4] = {l[0] + l[1] + l[2] + l[3], l[0] - l[1] + l[2] - l[3], l[0] + l[1] - l[2] - l[3], l[0] - l[1] - l[2] + l[3]};
    for (int i = 0; i < 4; i++) dc[i] = ((f[i] * ls00) * (1 << (qpc / 6))) >> 5;
}

/* ------------------------------------------------------------------ least-squares quantisation */
/* 1-D basis vectors of the inverse transforms in floating point (">> 1" taken as exact halves):
 * basis[u][x] = response at x of a unit coefficient u. */
static double basis4[4][4], norm4[4], basis8[8][8], norm8[8];
static int basis_ready;
static void finv4(double *p) {
    double a = p[0], b = p[1], c = p[2], d = p[3];
    double s0 = a + c, s1 = a - c, s2 = b / 2 - d, s3 = b + d / 2;
    p[0] = s0 + s3, p[1] = s1 + s2, p[2] = s1 - s2, p[3] = s0 - s3;
}
static void finv8(double *p) {
    double d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4], d5 = p[5], d6 = p[6], d7 = p[7];
    double a0 = d0 + d4, a2 = d0 - d4, a4 = d2 / 2 - d6, a6 = d2 + d6 / 2;
    double a1 = d5 - d3 - d7 - d7 / 2, a3 = d1 + d7 - d3 - d3 / 2, a5 = d7 - d1 + d5 + d5 / 2, a7 = d3 + d5 + d1 + d1 / 2;
    double b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    double b1 = a1 + a7 / 4, b3 = a3 + a5 / 4, b5 = a3 / 4 - a5, b7 = a7 - a1 / 4;
    p[0] = b0 + b7, p[1] = b2 + b5, p[2] = b4 + b3, p[3] = b6 + b1, p[4] = b6 - b1, p[5] = b4 - b3, p[6] = b2 - b5, p[7] = b0 - b7;
}
static void init_basis(void) {
    if (basis_ready) return;
    for (int u = 0; u < 4; u++) {
        double v[4] = {0, 0, 0, 0};
        v[u] = 1;
        finv4(v);
        norm4[u] = 0;
        for (int x = 0; x < 4; x++) basis4[u][x] = v[x], norm4[u] += v[x] * v[x];
    }
    for (int u = 0; u < 8; u++) {
        double v[8] = {0};
        v[u] = 1;
        finv8(v);
        norm8[u] = 0;
        for (int x = 0; x < 8; x++) basis8[u][x] = v[x], norm8[u] += v[x] * v[x];
    }
    basis_ready = 1;
}
static int16_t dead_round(double v, double dead) {
    double a = fabs(v);
    int q = (int)floor(a + dead);
    if (q > 2000) q = 2000;
    return (int16_t)(v < 0 ? -q : q);
}
void sg_quant4(const int *resid, const int *ls, int qp, double dead, int skip_dc, int16_t *lev) {
    init_basis();
    double scale = ldexp(1.0, qp / 6 - 4);
    for (int k = 0; k < 16; k++) {
        int pos = g_scan4[k], u = pos & 3, v = pos >> 2;
        if (skip_dc && k == 0) {
            lev[0] = 0;
            continue;
        }
        double acc = 0;
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) acc += resid[y * 4 + x] * basis4[v][y] * basis4[u][x];
        /* resid ~= lev * ls * scale * B / 64 */
        lev[k] = dead_round(64.0 * acc / (norm4[u] * norm4[v] * ls[pos] * scale), dead);
    }
}
void sg_quant8(const int *resid, const int *ls, int qp, double dead, int16_t *lev) {
    init_basis();
    double scale = ldexp(1.0, qp / 6 - 6);
    for (int k = 0; k < 64; k++) {
        int pos = g_scan8[k], u = pos & 7, v = pos >> 3;
        double acc = 0;
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) acc += resid[y * 8 + x] * basis8[v][y] * basis8[u][x];
        lev[k] = dead_round(64.0 * acc / (norm8[u] * norm8[v] * ls[pos] * scale), dead);
    }
}
void sg_quant_luma_dc(const int *sums, int ls00, int qp, double dead, int16_t *lev) {
    /* wanted DC coefficient of block i: d_i = 64 * sum_i / 16; d = (H c H) * ls00 * 2^(qp/6-6)  =>  c = H d H / 16 / (...) */
    int d4[16], f[16];
    for (int i = 0; i < 16; i++) d4[i] = sums[i];
    hadamard4(d4, f);
    double scale = ls00 * ldexp(1.0, qp / 6 - 6);
    for (int k = 0; k < 16; k++) lev[k] = dead_round(4.0 * f[g_scan4[k]] / 16.0 / scale, dead);
}
void sg_quant_chroma_dc(const int *s, int ls00, int qpc, double dead, int16_t *lev) {
    int f[4] = {s[0] + s[1] + s[2] + s[3], s[0] - s[1] + s[2] - s[3], s[0] + s[1] - s[2] - s[3], s[0] - s[1] - s[2] + s[3]};
    double scale = ls00 * ldexp(1.0, qpc / 6) / 32.0;
    for (int i = 0; i < 4; i++) lev[i] = dead_round(4.0 * f[i] / 4.0 / scale, dead);
}

/* ------------------------------------------------------------------ deblocking 8.7 */
static void edge_line(uint8_t *q0p, int step, int bs, int alpha, int beta, int tc0, int is_chroma) {
    int p[4], q[4];
    for (int i = 0; i < 4; i++) {
        if (is_chroma && i >= 2) {
            p[i] = q[i] = 0;
            continue;
        }
        p[i] = q0p[-(i + 1) * step];
        q[i] = q0p[i * step];
    }
    if (abs(p[0] - q[0]) >= alpha || abs(p[1] - p[0]) >= beta || abs(q[1] - q[0]) >= beta) return;
    int np[3] = {p[0], p[1], p[2]}, nq[3] = {q[0], q[1], q[2]};
    if (bs == 4) {
        if (is_chroma) {
            np[0] = (2 * p[1] + p[0] + q[1] + 2) >> 2;
            nq[0] = (2 * q[1] + q[0] + p[1] + 2) >> 2;
        } else {
            int strong = abs(p[0] - q[0]) < (alpha >> 2) + 2;
            if (strong && abs(p[2] - p[0]) < beta) {
                np[0] = (p[2] + 2 * p[1] + 2 * p[0] + 2 * q[0] + q[1] + 4) >> 3;
                np[1] = (p[2] + p[1] + p[0] + q[0] + 2) >> 2;
                np[2] = (2 * p[3] + 3 * p[2] + p[1] + p[0] + q[0] + 4) >> 3;
            } else
                np[0] = (2 * p[1] + p[0] + q[1] + 2) >> 2;
            if (strong && abs(q[2] - q[0]) < beta) {
                nq[0] = (q[2] + 2 * q[1] + 2 * q[0] + 2 * p[0] + p[1] + 4) >> 3;
                nq[1] = (q[2] + q[1] + q[0] + p[0] + 2) >> 2;
                nq[2] = (2 * q[3] + 3 * q[2] + q[1] + q[0] + p[0] + 4) >> 3;
            } else
                nq[0] = (2 * q[1] + q[0] + p[1] + 2) >> 2;
        }
    } else {
        int tc = tc0;
        if (is_chroma)
            tc += 1;
        else {
            if (abs(p[2] - p[0]) < beta) {
                tc++;
                np[1] = p[1] + iclip((p[2] + ((p[0] + q[0] + 1) >> 1) - 2 * p[1]) >> 1, -tc0, tc0);
            }
            if (abs(q[2] - q[0]) < beta) {
                tc++;
                nq[1] = q[1] + iclip((q[2] + ((p[0] + q[0] + 1) >> 1) - 2 * q[1]) >> 1, -tc0, tc0);
            }
        }
        int dlt = iclip((4 * (q[0] - p[0]) + (p[1] - q[1]) + 4) >> 3, -tc, tc);
        np[0] = clip255(p[0] + dlt);
        nq[0] = clip255(q[0] - dlt);
    }
    int n = is_chroma ? 1 : 3;
    for (int i = 0; i < n; i++) {
        q0p[-(i + 1) * step] = (uint8_t)np[i];
        q0p[i * step] = (uint8_t)nq[i];
    }
}
/* motion of one 4x4 block as a sorted set of (picture, vector) pairs: which list a picture came through is irrelevant (8.7.2.1) */
typedef struct {
    int n, pic[2], mvx[2], mvy[2];
} blkmotion;
static blkmotion motion_of(const sg_dbmb *m, int b) {
    blkmotion r;
    int q = (b / 8) * 2 + (b % 4) / 2;
    r.n = 0;
    if (m->refid[q] >= 0) r.pic[r.n] = m->refid[q], r.mvx[r.n] = m->mv[b][0], r.mvy[r.n] = m->mv[b][1], r.n++;
    if (m->refid1[q] >= 0) r.pic[r.n] = m->refid1[q], r.mvx[r.n] = m->mv1[b][0], r.mvy[r.n] = m->mv1[b][1], r.n++;
    return r;
}
static int close_mv(const blkmotion *a, int i, const blkmotion *b, int j) { return abs(a->mvx[i] - b->mvx[j]) < 4 && abs(a->mvy[i] - b->mvy[j]) < (g_field ? 2 : 4); }
static int strength(const sg_dbmb *mp, int bp, const sg_dbmb *mq, int bq, int on_mb_edge, int vertical_edge) {
    if (mp->intra || mq->intra) return on_mb_edge && (vertical_edge || !g_field) ? 4 : 3;
    if ((mp->nzmask >> bp & 1) || (mq->nzmask >> bq & 1)) return 2;
    blkmotion P = motion_of(mp, bp), Q = motion_of(mq, bq);
    if (P.n != Q.n) return 1;
    if (P.n == 1) return !(P.pic[0] == Q.pic[0] && close_mv(&P, 0, &Q, 0));
    /* two vectors each: the same two pictures, and some pairing of vectors into the same picture that is close */
    int straight = P.pic[0] == Q.pic[0] && P.pic[1] == Q.pic[1], crossed = P.pic[0] == Q.pic[1] && P.pic[1] == Q.pic[0];
    if (!straight && !crossed) return 1;
    if (straight && close_mv(&P, 0, &Q, 0) && close_mv(&P, 1, &Q, 1)) return 0;
    if (crossed && close_mv(&P, 0, &Q, 1) && close_mv(&P, 1, &Q, 0)) return 0;
    return 1;
}
void sg_deblock(sg_pic *pic, const sg_dbmb *mbs, int wmb, int hmb) {
    for (int my = 0; my < hmb; my++)
        for (int mx = 0; mx < wmb; mx++) {
            const sg_dbmb *cur = &mbs[my * wmb + mx];
            if (cur->dbf_idc == 1) continue;
            for (int vertical_edges = 1; vertical_edges >= 0; vertical_edges--) {
                const sg_dbmb *nb = NULL;
                if (vertical_edges && mx > 0) nb = cur - 1;
                if (!vertical_edges && my > 0) nb = cur - wmb;
                if (nb && cur->dbf_idc == 2 && nb->slice_id != cur->slice_id) nb = NULL;
                for (int e = 0; e < 4; e++) {
                    const sg_dbmb *pm = e ? cur : nb;
                    if (!pm) continue;
                    if (cur->t8x8 && (e & 1)) continue;
                    int bs[4], any = 0;
                    for (int s = 0; s < 4; s++) {
                        int bq = vertical_edges ? 4 * s + e : 4 * e + s;
                        int bp = vertical_edges ? 4 * s + (e ? e - 1 : 3) : 4 * (e ? e - 1 : 3) + s;
                        bs[s] = strength(pm, bp, cur, bq, e == 0, vertical_edges);
                        any |= bs[s];
                    }
                    if (!any) continue;
                    for (int plane = 0; plane < 3; plane++) {
                        if (plane && (e & 1)) break;
                        int stride = plane ? pic->w / 2 : pic->w, mbsz = plane ? 8 : 16;
                        int qa = plane ? (pm->qpc[plane - 1] + cur->qpc[plane - 1] + 1) >> 1 : (pm->qp + cur->qp + 1) >> 1;
                        int ia = iclip(qa + cur->alpha_off, 0, 51), ib = iclip(qa + cur->beta_off, 0, 51);
                        uint8_t *org = pic->pl[plane] + my * mbsz * stride + mx * mbsz;
                        int epos = plane ? e * 2 : e * 4;
                        for (int i = 0; i < mbsz; i++) {
                            int b = bs[plane ? i >> 1 : i >> 2];
                            if (!b) continue;
                            uint8_t *q0 = vertical_edges ? org + i * stride + epos : org + epos * stride + i;
                            edge_line(q0, vertical_edges ? 1 : stride, b, sg_alpha[ia], sg_beta[ib], b < 4 ? sg_tc0[ia][b - 1] : 0, plane != 0);
                        }
                    }
                }
            }
        }
}

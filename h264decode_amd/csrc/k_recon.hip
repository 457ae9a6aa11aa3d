// h264decode_amd/csrc/k_recon.hip -- K3 (intra reconstruction) and K6 (crop + pack), gfx950.  (K4 is k_inter.hip.)
//
// Everything here is byte / int16 work bounded by HBM traffic and LDS latency (no MFMA):
//   residual: scaling (8.5.9, 8.5.12.1) + Intra16x16 / chroma DC transforms (8.5.10, 8.5.11) +
//             4x4 / 8x8 inverse transforms (8.5.12.2, 8.5.13), two LDS passes (rows, columns);
//   K3 intra: macroblocks depend on their left / top / top-right neighbours, so a picture is
//             decoded by ONE workgroup (no cross-CU visibility problem): wavefront w owns
//             macroblock rows w, w+16, ..., and waits on an LDS progress counter of the row above
//             (2-D wavefront order).  Inside a macroblock the 4x4 / 8x8 blocks are reconstructed in
//             an LDS tile, so the serial block-to-block dependency never touches HBM.
//
// The reference has none of this (README.md:10 "Macroblock to YCbCr image decoding" is a TODO);
// normative source: ITU-T H.264 8.3, 8.4.2, 8.5.
#include <hip/hip_runtime.h>
#include <cstddef>
#include "mi_kernels.h"

#define WAVE_SYNC()                                             \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)

// Frame memory through explicit address-space-1 pointers: the frame pool's address arrives as an integer (PicDesc::pool_base),
// and a pointer made from an integer is generic -- flat_load / flat_store, which are slower and count against the LDS wait
// counter as well.
typedef __attribute__((address_space(1))) uint8_t g8;
typedef __attribute__((address_space(1))) uint16_t g16;
typedef __attribute__((address_space(1))) uint32_t g32;

__device__ __forceinline__ int clip255(int v) { return min(max(v, 0), 255); }

struct ResBuf {
    int32_t tmp[384];      // row-pass output: luma [0..255], chroma [256..383]
    int16_t luma[256];     // residual, raster 16x16
    int16_t chroma[2][64]; // residual, raster 8x8 per plane
    int32_t dc[24];        // Intra16x16 DC (16, block raster) + chroma DC (2 x 4)
};

// ------------------------------------------------------------------ 1-D inverse transforms
__device__ __forceinline__ void inv4(int d0, int d1, int d2, int d3, int &o0, int &o1, int &o2, int &o3) {
    int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);
    o0 = e0 + e3, o1 = e1 + e2, o2 = e1 - e2, o3 = e0 - e3;
}
__device__ __forceinline__ void inv8(const int *d, int *o) {
    int e0 = d[0] + d[4], e1 = -d[3] + d[5] - d[7] - (d[7] >> 1), e2 = d[0] - d[4], e3 = d[1] + d[7] - d[3] - (d[3] >> 1);
    int e4 = (d[2] >> 1) - d[6], e5 = -d[1] + d[7] + d[5] + (d[5] >> 1), e6 = d[2] + (d[6] >> 1), e7 = d[3] + d[5] + d[1] + (d[1] >> 1);
    int f0 = e0 + e6, f1 = e1 + (e7 >> 2), f2 = e2 + e4, f3 = e3 + (e5 >> 2);
    int f4 = e2 - e4, f5 = (e3 >> 2) - e5, f6 = e0 - e6, f7 = e7 - (e1 >> 2);
    o[0] = f0 + f7, o[1] = f2 + f5, o[2] = f4 + f3, o[3] = f6 + f1;
    o[4] = f6 - f1, o[5] = f4 - f3, o[6] = f2 - f5, o[7] = f0 - f7;
}
__device__ __forceinline__ int scale4(int c, int ls, int qp) { // 8.5.12.1
    int per = qp / 6;
    return per >= 4 ? (c * ls) << (per - 4) : (c * ls + (1 << (3 - per))) >> (4 - per);
}
__device__ __forceinline__ int scale8(int c, int ls, int qp) { // 8.5.13 scaling
    int per = qp / 6;
    return per >= 6 ? (c * ls) << (per - 6) : (c * ls + (1 << (5 - per))) >> (6 - per);
}

// Residual of one macroblock, computed by one wavefront (lane = 0..63).  Results in rb->luma / rb->chroma.
__device__ __forceinline__ void mb_residual(int lane, const MbRec *rec, const int16_t *coef, const ScalingSet *sc, ResBuf *rb) {
    const int type = rec->type, t8x8 = rec->t8x8, cbp = rec->cbp, qp = rec->qp;
    const int intra = MB_IS_INTRA(type), i16 = type == MBT_I16x16;
    const int cbp_l = cbp & 15, cbp_c = cbp >> 4;
    // ---- DC transforms ----
    if (i16 && lane < 16) { // 8.5.10: f = A c A, A = 4x4 Hadamard
        const int i = lane >> 2, j = lane & 3;
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // A[i][k]: row i of the Hadamard matrix
            int aik = (i == 0) ? 1 : (i == 1 ? (k < 2 ? 1 : -1) : (i == 2 ? ((k == 0 || k == 3) ? 1 : -1) : ((k & 1) ? -1 : 1)));
            int rowsum = 0;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                int amj = (j == 0) ? 1 : (j == 1 ? (m < 2 ? 1 : -1) : (j == 2 ? ((m == 0 || m == 3) ? 1 : -1) : ((m & 1) ? -1 : 1)));
                rowsum += coef[MI_COEF_I16DC + k * 4 + m] * amj;
            }
            acc += aik * rowsum;
        }
        int ls00 = sc->ls4[0][qp % 6][0], per = qp / 6;
        rb->dc[lane] = per >= 6 ? (acc * ls00) << (per - 6) : (acc * ls00 + (1 << (5 - per))) >> (6 - per);
    }
    if (cbp_c && lane >= 16 && lane < 24) { // 8.5.11
        const int c = (lane - 16) >> 2, b = lane & 3;
        const int16_t *p = coef + MI_COEF_CDC + 4 * c;
        int c0 = p[0], c1 = p[1], c2 = p[2], c3 = p[3];
        int f = b == 0 ? c0 + c1 + c2 + c3 : (b == 1 ? c0 - c1 + c2 - c3 : (b == 2 ? c0 + c1 - c2 - c3 : c0 - c1 - c2 + c3));
        int qpc = rec->qpc[c], ls00 = sc->ls4[(intra ? 1 : 4) + c][qpc % 6][0];
        rb->dc[16 + c * 4 + b] = ((f * ls00) << (qpc / 6)) >> 5;
    }
    WAVE_SYNC();
    // ---- row pass ----
    if (t8x8) {
        if (lane < 32) {
            const int b8 = lane >> 3, row = lane & 7;
            int o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if ((cbp_l >> b8) & 1) {
                const uint16_t *ls = sc->ls8[intra ? 0 : 1][qp % 6] + row * 8;
                const int16_t *c = coef + b8 * 64 + row * 8;
                int d[8];
#pragma unroll
                for (int k = 0; k < 8; k++) d[k] = scale8(c[k], ls[k], qp);
                inv8(d, o);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) rb->tmp[b8 * 64 + row * 8 + k] = o[k];
        }
    } else {
        const int r = lane >> 2, row = lane & 3, b8 = ((r >> 3) << 1) | ((r & 3) >> 1);
        int o0 = 0, o1 = 0, o2 = 0, o3 = 0;
        const int coded = (cbp_l >> b8) & 1;
        if (coded || i16) {
            const uint16_t *ls = sc->ls4[intra ? 0 : 3][qp % 6] + row * 4;
            const int16_t *c = coef + r * 16 + row * 4;
            int d0 = coded ? scale4(c[0], ls[0], qp) : 0, d1 = coded ? scale4(c[1], ls[1], qp) : 0;
            int d2 = coded ? scale4(c[2], ls[2], qp) : 0, d3 = coded ? scale4(c[3], ls[3], qp) : 0;
            if (i16 && row == 0) d0 = rb->dc[r];
            inv4(d0, d1, d2, d3, o0, o1, o2, o3);
        }
        int32_t *t = rb->tmp + r * 16 + row * 4;
        t[0] = o0, t[1] = o1, t[2] = o2, t[3] = o3;
    }
    if (cbp_c && lane < 32) { // chroma rows: 2 planes x 4 blocks x 4 rows (no coded chroma: the column pass writes zeros)
        const int c = lane >> 4, b = (lane >> 2) & 3, row = lane & 3;
        int o0 = 0, o1 = 0, o2 = 0, o3 = 0;
        {
            const int qpc = rec->qpc[c];
            const uint16_t *ls = sc->ls4[(intra ? 1 : 4) + c][qpc % 6] + row * 4;
            const int16_t *p = coef + MI_COEF_CAC + (c * 4 + b) * 16 + row * 4;
            const int ac = cbp_c & 2;
            int d0 = ac ? scale4(p[0], ls[0], qpc) : 0, d1 = ac ? scale4(p[1], ls[1], qpc) : 0;
            int d2 = ac ? scale4(p[2], ls[2], qpc) : 0, d3 = ac ? scale4(p[3], ls[3], qpc) : 0;
            if (row == 0) d0 = rb->dc[16 + c * 4 + b];
            inv4(d0, d1, d2, d3, o0, o1, o2, o3);
        }
        int32_t *t = rb->tmp + 256 + (c * 4 + b) * 16 + row * 4;
        t[0] = o0, t[1] = o1, t[2] = o2, t[3] = o3;
    }
    WAVE_SYNC();
    // ---- column pass ----
    if (t8x8) {
        if (lane < 32) {
            const int b8 = lane >> 3, col = lane & 7;
            int d[8], o[8];
#pragma unroll
            for (int k = 0; k < 8; k++) d[k] = rb->tmp[b8 * 64 + k * 8 + col];
            inv8(d, o);
            const int x0 = (b8 & 1) * 8 + col, y0 = (b8 >> 1) * 8;
#pragma unroll
            for (int k = 0; k < 8; k++) rb->luma[(y0 + k) * 16 + x0] = static_cast<int16_t>((o[k] + 32) >> 6);
        }
    } else {
        const int r = lane >> 2, col = lane & 3;
        const int32_t *t = rb->tmp + r * 16 + col;
        int o0, o1, o2, o3;
        inv4(t[0], t[4], t[8], t[12], o0, o1, o2, o3);
        const int x0 = (r & 3) * 4 + col, y0 = (r >> 2) * 4;
        rb->luma[(y0 + 0) * 16 + x0] = static_cast<int16_t>((o0 + 32) >> 6);
        rb->luma[(y0 + 1) * 16 + x0] = static_cast<int16_t>((o1 + 32) >> 6);
        rb->luma[(y0 + 2) * 16 + x0] = static_cast<int16_t>((o2 + 32) >> 6);
        rb->luma[(y0 + 3) * 16 + x0] = static_cast<int16_t>((o3 + 32) >> 6);
    }
    if (!cbp_c)
        reinterpret_cast<uint32_t *>(rb->chroma)[lane] = 0u; // 2 x 64 int16
    else if (lane < 32) {
        const int c = lane >> 4, b = (lane >> 2) & 3, col = lane & 3;
        const int32_t *t = rb->tmp + 256 + (c * 4 + b) * 16 + col;
        int o0, o1, o2, o3;
        inv4(t[0], t[4], t[8], t[12], o0, o1, o2, o3);
        const int x0 = (b & 1) * 4 + col, y0 = (b >> 1) * 4;
        int16_t *dst = rb->chroma[c];
        dst[(y0 + 0) * 8 + x0] = static_cast<int16_t>((o0 + 32) >> 6);
        dst[(y0 + 1) * 8 + x0] = static_cast<int16_t>((o1 + 32) >> 6);
        dst[(y0 + 2) * 8 + x0] = static_cast<int16_t>((o2 + 32) >> 6);
        dst[(y0 + 3) * 8 + x0] = static_cast<int16_t>((o3 + 32) >> 6);
    }
    WAVE_SYNC();
}

__device__ __forceinline__ void zero_residual(int lane, ResBuf *rb) {
    for (int i = lane; i < 128; i += 64) reinterpret_cast<uint32_t *>(rb->luma)[i] = 0;
    reinterpret_cast<uint32_t *>(rb->chroma)[lane] = 0;
    WAVE_SYNC();
}

// ================================================================== K3: intra prediction
struct IntraWave {
    ResBuf rb;
    uint8_t tile[17][28];     // luma: row 0 = samples above, column 0 = samples to the left; 24 columns to the right for top-right
    uint8_t tile_c[2][9][12]; // chroma
    int16_t fe[2][32];        // Intra8x8 filtered reference samples: [0] top p'[-1..15] at index x+1, [1] left p'[-1..7] at index y+1
    MbRec rec;                // LDS copy of the current macroblock record
    alignas(16) int16_t coef[MI_COEF_PER_MB]; // dense coefficient layout of the macroblock, scattered from the packed pool
};
#define MI_INTRA_MAX_ROWS 320  /* macroblock rows (5120 luma lines) */
#define MI_INTRA_MAX_CHUNKS 8 /* 64-macroblock chunks per row (8192 luma columns) */
struct IntraShared {
    IntraWave w[MI_INTRA_WAVES];
    ScalingSet sc; // LevelScale tables of the picture
    // intra macroblocks still to be reconstructed, one bit each: an intra macroblock waits for exactly the intra
    // macroblocks among its upper-left / upper / upper-right neighbours (inter ones were finished by K4), so the
    // isolated intra macroblocks of P pictures do not serialise behind each other row after row
    unsigned long long pend[MI_INTRA_MAX_ROWS][MI_INTRA_MAX_CHUNKS];
};

// directional Intra4x4 / Intra8x8 predictors (8.3.1.2.4-9, 8.3.2.2.5-10).  T(x) = p[x,-1], L(y) = p[-1,y], T(-1)=L(-1)=p[-1,-1]
template <int N, typename FT, typename FL>
__device__ __forceinline__ int pred_dir(int mode, int x, int y, FT T, FL L) {
    switch (mode) {
    case 3: return (x == N - 1 && y == N - 1) ? (T(2 * N - 2) + 3 * T(2 * N - 1) + 2) >> 2 : (T(x + y) + 2 * T(x + y + 1) + T(x + y + 2) + 2) >> 2;
    case 4:
        if (x > y) return (T(x - y - 2) + 2 * T(x - y - 1) + T(x - y) + 2) >> 2;
        if (x < y) return (L(y - x - 2) + 2 * L(y - x - 1) + L(y - x) + 2) >> 2;
        return (T(0) + 2 * T(-1) + L(0) + 2) >> 2;
    case 5: {
        int z = 2 * x - y, k = x - (y >> 1);
        if (z < -1) return (L(y - 2 * x - 1) + 2 * L(y - 2 * x - 2) + L(y - 2 * x - 3) + 2) >> 2;
        if (z == -1) return (L(0) + 2 * T(-1) + T(0) + 2) >> 2;
        if (!(z & 1)) return (T(k - 1) + T(k) + 1) >> 1;
        return (T(k - 2) + 2 * T(k - 1) + T(k) + 2) >> 2;
    }
    case 6: {
        int z = 2 * y - x, k = y - (x >> 1);
        if (z < -1) return (T(x - 2 * y - 1) + 2 * T(x - 2 * y - 2) + T(x - 2 * y - 3) + 2) >> 2;
        if (z == -1) return (L(0) + 2 * T(-1) + T(0) + 2) >> 2;
        if (!(z & 1)) return (L(k - 1) + L(k) + 1) >> 1;
        return (L(k - 2) + 2 * L(k - 1) + L(k) + 2) >> 2;
    }
    case 7: {
        int k = x + (y >> 1);
        return !(y & 1) ? (T(k) + T(k + 1) + 1) >> 1 : (T(k) + 2 * T(k + 1) + T(k + 2) + 2) >> 2;
    }
    default: {
        int z = x + 2 * y, k = y + (x >> 1);
        if (z > 2 * N - 3) return L(N - 1);
        if (z == 2 * N - 3) return (L(N - 2) + 3 * L(N - 1) + 2) >> 2;
        if (!(z & 1)) return (L(k) + L(k + 1) + 1) >> 1;
        return (L(k) + 2 * L(k + 1) + L(k + 2) + 2) >> 2;
    }
    }
}

// Intra16x16 (N=16) and chroma (N=8) plane prediction; T/L as above
template <int N, typename FT, typename FL>
__device__ __forceinline__ void plane_params(FT T, FL L, int &a, int &b, int &c) {
    int hh = 0, vv = 0;
    constexpr int m = N / 2;
#pragma unroll
    for (int k = 1; k <= m; k++) {
        hh += k * (T(m - 1 + k) - T(m - 1 - k));
        vv += k * (L(m - 1 + k) - L(m - 1 - k));
    }
    a = 16 * (L(N - 1) + T(N - 1));
    b = N == 16 ? (5 * hh + 32) >> 6 : (34 * hh + 32) >> 6;
    c = N == 16 ? (5 * vv + 32) >> 6 : (34 * vv + 32) >> 6;
}

// Frame accesses of K3.  X = true (k_intra_x: a picture spread over several workgroups): the samples a macroblock leaves are
// read by macroblocks of OTHER CUs in the same launch, so every store is an agent-scope (sc1, write-through) store and every
// neighbour sample is read by an agent-scope load (past this CU's L1, which other CUs' stores never refresh).
template <bool X>
__device__ __forceinline__ int ld_px(const g8 *p) {
    if (X) return static_cast<int>(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return static_cast<int>(*p);
}
template <bool X>
__device__ __forceinline__ void st_px32(g8 *p, uint32_t v) {
    if (X)
        __hip_atomic_store(reinterpret_cast<g32 *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *reinterpret_cast<g32 *>(p) = v;
}
template <bool X>
__device__ __forceinline__ void st_px16(g8 *p, uint16_t v) {
    if (X)
        __hip_atomic_store(reinterpret_cast<g16 *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *reinterpret_cast<g16 *>(p) = v;
}

template <bool X>
__device__ void intra_mb(int lane, IntraWave *ws, const MbRec *rec, const int16_t *coef, const ScalingSet *sc, g8 *py, g8 *pcb, g8 *pcr, int W,
                         int mbx, int mby) {
    const int type = rec->type, av = rec->avail;
    const int a_left = av & MI_AV_LEFT, a_top = (av & MI_AV_TOP) != 0, a_tl = (av & MI_AV_TOPLEFT) != 0, a_tr = (av & MI_AV_TOPRIGHT) != 0;
    // (frame pointers carry address space 1: built from an integer they would otherwise be generic -- flat_load / flat_store,
    // which count against the LDS wait counter too, so that every LDS fence below would wait for the sample stores)
    g8 *Y = py + static_cast<size_t>(mby * 16) * W + mbx * 16;
    const int Wc = W / 2;
    g8 *C0 = pcb + static_cast<size_t>(mby * 8) * Wc + mbx * 8, *C1 = pcr + static_cast<size_t>(mby * 8) * Wc + mbx * 8;
#define CPL(c) ((c) ? C1 : C0) /* a select, not an indexed pointer array (which would live in scratch) */
    if (type == MBT_NONE) { // lost macroblock: a defined background instead of whatever the slot held before
        st_px32<X>(Y + static_cast<size_t>(lane >> 2) * W + (lane & 3) * 4, 0x80808080u);
        if (lane < 32) st_px32<X>(CPL(lane >> 4) + static_cast<size_t>((lane >> 1) & 7) * Wc + (lane & 1) * 4, 0x80808080u);
        return;
    }
    if (type == MBT_IPCM) { // 8.3.5: samples were stored in the coefficient block
        const uint8_t *pcm = reinterpret_cast<const uint8_t *>(coef);
        { // 256 luma bytes: one dword per lane
            int j = lane >> 2, i = (lane & 3) * 4;
            st_px32<X>(Y + static_cast<size_t>(j) * W + i, *reinterpret_cast<const uint32_t *>(pcm + j * 16 + i));
        }
        if (lane < 32) {
            int c = lane >> 4, j = (lane >> 1) & 7, i = (lane & 1) * 4;
            st_px32<X>(CPL(c) + static_cast<size_t>(j) * Wc + i, *reinterpret_cast<const uint32_t *>(pcm + 256 + c * 64 + j * 8 + i));
        }
        return;
    }
    // ---- neighbouring samples into the LDS tiles ----
    if (lane < 25) { // row above: x = -1 .. 23
        int x = lane - 1;
        int ok = x < 0 ? a_tl : (x < 16 ? a_top : a_tr);
        ws->tile[0][lane] = ok ? static_cast<uint8_t>(ld_px<X>(Y - static_cast<ptrdiff_t>(W) + x)) : static_cast<uint8_t>(128);
    } else if (lane >= 32 && lane < 48) {
        int y = lane - 32;
        ws->tile[y + 1][0] = a_left ? static_cast<uint8_t>(ld_px<X>(Y + static_cast<size_t>(y) * W - 1)) : static_cast<uint8_t>(128);
    }
    if (lane < 18) { // chroma rows above: x = -1..7 for both planes
        int c = lane / 9, x = lane % 9 - 1;
        int ok = x < 0 ? a_tl : a_top;
        ws->tile_c[c][0][x + 1] = ok ? static_cast<uint8_t>(ld_px<X>(CPL(c) - static_cast<ptrdiff_t>(Wc) + x)) : static_cast<uint8_t>(128);
    } else if (lane >= 32 && lane < 48) {
        int c = (lane - 32) >> 3, y = lane & 7;
        ws->tile_c[c][y + 1][0] = a_left ? static_cast<uint8_t>(ld_px<X>(CPL(c) + static_cast<size_t>(y) * Wc - 1)) : static_cast<uint8_t>(128);
    }
    // ---- residual ----
    if ((rec->cbp & 0x3F) || type == MBT_I16x16)
        mb_residual(lane, rec, coef, sc, &ws->rb);
    else
        zero_residual(lane, &ws->rb);
    // ---- luma ----
    if (type == MBT_I16x16) {
        const int mode = rec->i16mode;
        auto T = [&](int x) { return static_cast<int>(ws->tile[0][x + 1]); };
        auto L = [&](int y) { return static_cast<int>(ws->tile[y + 1][0]); };
        const int j = lane >> 2, i0 = (lane & 3) * 4;
        int dc = 128, pa = 0, pb = 0, pc = 0;
        if (mode == 2) {
            int st = 0, sl = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) st += T(k), sl += L(k);
            if (a_top && a_left)
                dc = (st + sl + 16) >> 5;
            else if (a_top)
                dc = (st + 8) >> 4;
            else if (a_left)
                dc = (sl + 8) >> 4;
        } else if (mode == 3)
            plane_params<16>(T, L, pa, pb, pc);
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int x = i0 + k, v;
            if (mode == 0)
                v = T(x);
            else if (mode == 1)
                v = L(j);
            else if (mode == 2)
                v = dc;
            else
                v = clip255((pa + pb * (x - 7) + pc * (j - 7) + 16) >> 5);
            v = clip255(v + ws->rb.luma[j * 16 + x]);
            packed |= static_cast<uint32_t>(v) << (8 * k);
        }
        st_px32<X>(Y + static_cast<size_t>(j) * W + i0, packed);
    } else if (type == MBT_I4x4) {
        // The sixteen 4x4 blocks in ten steps instead of sixteen: a block predicts from its left, upper-left, upper and upper-right neighbours, so
        // the blocks on an anti-diagonal x + 2y = d are independent -- step d reconstructs up to two of them side by side (lanes 0..15 the upper
        // one, 16..31 the lower one).  Which neighbours count as available stays what the z-order of 6.4.3 makes it (an upper-right block
        // that comes later in that order is "not available" even though this schedule has already reconstructed it).
        for (int d = 0; d < 10; d++) {
            const int half = lane >> 4;                                   // 0: the block with the larger x (or the only one), 1: the one a row further down
            const int by = (d >= 2 && d <= 7) ? (d & 1 ? (half ? 1 + ((d - 3) >> 1) : (d - 3) >> 1) : (half ? (d >> 1) : (d >> 1) - 1)) : (d < 2 ? 0 : 3);
            const int bx = d - 2 * by;
            const bool two = d >= 2 && d <= 7;
            if (lane < (two ? 32 : 16)) {
                const int mode = rec->ipm[by * 4 + bx];
                const int has_left = bx > 0 || a_left, has_top = by > 0 || a_top;
                int has_tr;
                if (by == 0)
                    has_tr = bx < 3 ? a_top : a_tr;
                else
                    has_tr = bx == 3 ? 0 : !((bx & 1) && (by & 1));
                const int x = lane & 3, y = (lane >> 2) & 3;
                const uint8_t *trow = &ws->tile[by * 4][bx * 4 + 1]; // trow[x] = p[x,-1], trow[-1] = p[-1,-1]
                auto T = [&](int k) { return static_cast<int>((k >= 4 && !has_tr) ? trow[3] : trow[k]); };
                auto L = [&](int k) { return k < 0 ? static_cast<int>(trow[-1]) : static_cast<int>(ws->tile[by * 4 + 1 + k][bx * 4]); };
                int v;
                if (mode == 0)
                    v = T(x);
                else if (mode == 1)
                    v = L(y);
                else if (mode == 2) {
                    int st = T(0) + T(1) + T(2) + T(3), sl = L(0) + L(1) + L(2) + L(3);
                    v = (has_top && has_left) ? (st + sl + 4) >> 3 : (has_top ? (st + 2) >> 2 : (has_left ? (sl + 2) >> 2 : 128));
                } else
                    v = pred_dir<4>(mode, x, y, T, L);
                v = clip255(v + ws->rb.luma[(by * 4 + y) * 16 + bx * 4 + x]);
                ws->tile[by * 4 + 1 + y][bx * 4 + 1 + x] = static_cast<uint8_t>(v);
            }
            WAVE_SYNC();
        }
    } else { // MBT_I8x8
        for (int b8 = 0; b8 < 4; b8++) {
            const int x8 = b8 & 1, y8 = b8 >> 1;
            const int mode = rec->ipm[y8 * 8 + x8 * 2];
            const int has_left = x8 || a_left, has_top = y8 || a_top;
            const int has_tl = b8 == 0 ? a_tl : (b8 == 1 ? a_top : (b8 == 2 ? (a_left != 0) : 1));
            const int has_tr = b8 == 0 ? a_top : (b8 == 1 ? a_tr : (b8 == 2 ? 1 : 0));
            const uint8_t *trow = &ws->tile[y8 * 8][x8 * 8 + 1];
            auto Tr = [&](int k) { return k < 0 ? static_cast<int>(trow[-1]) : static_cast<int>((k >= 8 && !has_tr) ? trow[7] : trow[k]); };
            auto Lr = [&](int k) { return k < 0 ? static_cast<int>(trow[-1]) : static_cast<int>(ws->tile[y8 * 8 + 1 + k][x8 * 8]); };
            // 8.3.2.2.1 reference sample filtering, one sample per lane
            if (lane < 17) {
                int x = lane - 1, v;
                if (x < 0) {
                    if (has_top && has_left)
                        v = (Tr(0) + 2 * Tr(-1) + Lr(0) + 2) >> 2;
                    else if (has_top)
                        v = (3 * Tr(-1) + Tr(0) + 2) >> 2;
                    else if (has_left)
                        v = (3 * Tr(-1) + Lr(0) + 2) >> 2;
                    else
                        v = Tr(-1);
                } else if (x == 0)
                    v = has_tl ? (Tr(-1) + 2 * Tr(0) + Tr(1) + 2) >> 2 : (3 * Tr(0) + Tr(1) + 2) >> 2;
                else if (x < 15)
                    v = (Tr(x - 1) + 2 * Tr(x) + Tr(x + 1) + 2) >> 2;
                else
                    v = (Tr(14) + 3 * Tr(15) + 2) >> 2;
                ws->fe[0][lane] = static_cast<int16_t>(v);
            } else if (lane >= 32 && lane < 40) {
                int y = lane - 32, v;
                if (y == 0)
                    v = has_tl ? (Lr(-1) + 2 * Lr(0) + Lr(1) + 2) >> 2 : (3 * Lr(0) + Lr(1) + 2) >> 2;
                else if (y < 7)
                    v = (Lr(y - 1) + 2 * Lr(y) + Lr(y + 1) + 2) >> 2;
                else
                    v = (Lr(6) + 3 * Lr(7) + 2) >> 2;
                ws->fe[1][y + 1] = static_cast<int16_t>(v);
            }
            WAVE_SYNC();
            {
                const int x = lane & 7, y = lane >> 3;
                auto T = [&](int k) { return static_cast<int>(ws->fe[0][k + 1]); };
                auto L = [&](int k) { return k < 0 ? static_cast<int>(ws->fe[0][0]) : static_cast<int>(ws->fe[1][k + 1]); };
                int v;
                if (mode == 0)
                    v = T(x);
                else if (mode == 1)
                    v = L(y);
                else if (mode == 2) {
                    int st = 0, sl = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) st += T(k), sl += L(k);
                    v = (has_top && has_left) ? (st + sl + 8) >> 4 : (has_top ? (st + 4) >> 3 : (has_left ? (sl + 4) >> 3 : 128));
                } else
                    v = pred_dir<8>(mode, x, y, T, L);
                v = clip255(v + ws->rb.luma[(y8 * 8 + y) * 16 + x8 * 8 + x]);
                WAVE_SYNC(); // all lanes have read fe[] / tile before the block is written
                ws->tile[y8 * 8 + 1 + y][x8 * 8 + 1 + x] = static_cast<uint8_t>(v);
            }
            WAVE_SYNC();
        }
    }
    if (type != MBT_I16x16) { // tile -> frame: one dword per lane
        const int j = lane >> 2, i0 = (lane & 3) * 4;
        const uint8_t *t = &ws->tile[j + 1][i0 + 1];
        uint32_t packed = t[0] | (t[1] << 8) | (t[2] << 16) | (static_cast<uint32_t>(t[3]) << 24);
        st_px32<X>(Y + static_cast<size_t>(j) * W + i0, packed);
    }
    // ---- chroma (8.3.4) ----
    {
        const int mode = rec->chroma_mode;
        const int c = lane >> 5, q = lane & 31, y = q >> 2, x0 = (q & 3) * 2;
        auto T = [&](int k) { return static_cast<int>(ws->tile_c[c][0][k + 1]); };
        auto L = [&](int k) { return static_cast<int>(ws->tile_c[c][k + 1][0]); };
        int pa = 0, pb = 0, pc = 0, dc = 128;
        if (mode == 3)
            plane_params<8>(T, L, pa, pb, pc);
        else if (mode == 0) {
            const int xo = x0 & 4, yo = y & 4;
            int st = T(xo) + T(xo + 1) + T(xo + 2) + T(xo + 3), sl = L(yo) + L(yo + 1) + L(yo + 2) + L(yo + 3);
            int use_t = a_top, use_l = a_left != 0;
            if (xo && !yo && a_top)
                use_l = 0;
            else if (!xo && yo && a_left)
                use_t = 0;
            if (use_t && use_l)
                dc = (st + sl + 4) >> 3;
            else if (use_t)
                dc = (st + 2) >> 2;
            else if (use_l)
                dc = (sl + 2) >> 2;
        }
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            int x = x0 + k, v;
            if (mode == 0)
                v = dc;
            else if (mode == 1)
                v = L(y);
            else if (mode == 2)
                v = T(x);
            else
                v = clip255((pa + pb * (x - 3) + pc * (y - 3) + 16) >> 5);
            v = clip255(v + ws->rb.chroma[c][y * 8 + x]);
            packed |= static_cast<uint32_t>(v) << (8 * k);
        }
        st_px16<X>(CPL(c) + static_cast<size_t>(y) * Wc + x0, static_cast<uint16_t>(packed));
    }
}

// 64 bits of the batch's intra mask (k_dbprep: one bit per macroblock, index = position in the record array) from bit `first` on, the first `n` of them
__device__ __forceinline__ unsigned long long intra_bits(const unsigned long long *mask, unsigned long long first, int n) {
    const unsigned long long w = first >> 6;
    const int sh = static_cast<int>(first & 63);
    unsigned long long m = mask[w] >> sh;
    if (sh) m |= mask[w + 1] << (64 - sh);
    return n >= 64 ? m : (m & ((1ull << n) - 1ull));
}

extern "C" __global__ void __launch_bounds__(MI_INTRA_WAVES * 64) k_intra(const uint32_t *pic_list, const PicDesc *pics, const FramePool *pools, const DevTables *tab,
                                                                          const MbRec *mbrec, const int16_t *coefs, const unsigned long long *intramask) {
    __shared__ IntraShared sh;
    __shared__ unsigned long long s_intra[MI_INTRA_MAX_ROWS][MI_INTRA_MAX_CHUNKS]; // sh.pend as pass 1 left it: which macroblocks K3 owns
    __shared__ uint32_t s_prefix[MI_INTRA_MAX_ROWS + 1];                          // intra macroblocks in the rows before row r
    __shared__ uint32_t s_next;
    const int tid = static_cast<int>(threadIdx.x), lane = tid & 63, wave = tid >> 6;
    const PicDesc *pd = &pics[pic_list[blockIdx.x]];
    const int wmb = static_cast<int>(pd->wmb), hmb = static_cast<int>(pd->hmb);
    // the picture's place in its frame slot (PicDesc): W = bytes from one luma row of the PICTURE to the next -- twice the frame's
    // for a field picture, which lives in the rows of its parity
    const int W = static_cast<int>(pd->pitch);
    const uint32_t par_off = pd->field == 2 ? pd->pitch >> 1 : 0u;
    g8 *py = (g8 *)(pd->pool_base + static_cast<uint64_t>(pd->slot) * pd->slot_bytes) + par_off;
    g8 *pcb = py - par_off + pd->plane + (par_off >> 1), *pcr = pcb + (pd->plane >> 2);
    { // LevelScale tables of this picture's PPS -> LDS (2688 bytes)
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&tab->scaling[pd->scaling_set]);
        for (int i = tid; i < static_cast<int>(sizeof(ScalingSet) / 4); i += MI_INTRA_WAVES * 64) reinterpret_cast<uint32_t *>(&sh.sc)[i] = src[i];
    }
    const int nchunks = (wmb + 63) >> 6;
    const MbRec *recs = mbrec + pd->mb_base;
    // ---- pass 1: intra masks of every row, cut out of the batch's bit mask (k_dbprep; macroblocks no slice delivered -- type MBT_NONE -- are in
    // it too: the intra path paints them mid-grey).  One thread per (row, 64-macroblock chunk). ----
    for (int i = tid; i < hmb * nchunks; i += MI_INTRA_WAVES * 64) {
        const int mby = i / nchunks, c = i - mby * nchunks;
        const unsigned long long m = intra_bits(intramask, pd->mb_base + static_cast<unsigned long long>(mby) * wmb + c * 64, wmb - c * 64);
        sh.pend[mby][c] = m, s_intra[mby][c] = m;
    }
    if (tid == 0) s_next = 0;
    __syncthreads();
    if (tid == 0) { // running sums of the rows' counts (at most 320 rows)
        uint32_t acc = 0;
        s_prefix[0] = 0;
        for (int r = 0; r < hmb; r++) {
            for (int c = 0; c < nchunks; c++) acc += static_cast<uint32_t>(__builtin_popcountll(s_intra[r][c]));
            s_prefix[r + 1] = acc;
        }
    }
    __syncthreads();
    IntraWave *ws = &sh.w[wave];
    // ---- pass 2 ----
    // Dense pictures (I pictures): wavefront w owns rows w, w + MI_INTRA_WAVES, ...; left-to-right inside a row.
    // Sparse pictures (the intra macroblocks of P / B pictures): dealt out one by one in raster order from a counter.  Rows hold
    // very different numbers of them, so whole rows per wavefront leave most wavefronts idle behind the busiest; a macroblock only
    // ever waits for macroblocks before it in raster order -- handed out earlier, to wavefronts that are running -- so nobody
    // waits for work that has not started.  (Dense pictures would serialise on their left neighbours that way.)
    const uint32_t n_intra = s_prefix[hmb];
    const bool sparse = n_intra * 4u < static_cast<uint32_t>(wmb * hmb);
    int row = wave, chunk = -1; // dense: the (row, chunk) whose remaining macroblocks are in `mask`
    unsigned long long mask = 0;
    for (;;) { // (one loop, so that the macroblock body is inlined once: twice costs 200 spilled registers)
        int mbx, mby;
        if (sparse) {
            uint32_t n = 0;
            if (lane == 0) n = atomicAdd(&s_next, 1u);
            n = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(n)));
            if (n >= n_intra) break;
            int lo = 0, hi = hmb; // the row: prefix[lo] <= n < prefix[lo + 1]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_prefix[mid] <= n) lo = mid; else hi = mid;
            }
            uint32_t k = n - s_prefix[lo];
            int c = 0;
            for (; c < nchunks - 1; c++) {
                const uint32_t pc = static_cast<uint32_t>(__builtin_popcountll(s_intra[lo][c]));
                if (k < pc) break;
                k -= pc;
            }
            const unsigned long long m = s_intra[lo][c];
            const bool mine = ((m >> lane) & 1ull) && static_cast<uint32_t>(__builtin_popcountll(m & ((1ull << lane) - 1ull))) == k;
            mbx = c * 64 + __builtin_ctzll(__builtin_amdgcn_ballot_w64(mine) | (1ull << 63)), mby = lo;
        } else {
            while (!mask) {
                if (++chunk == nchunks) chunk = 0, row += MI_INTRA_WAVES;
                if (row >= hmb) break;
                mask = s_intra[row][chunk];
            }
            if (!mask) break;
            mbx = chunk * 64 + __ffsll(static_cast<long long>(mask)) - 1, mby = row;
            mask &= mask - 1;
        }
        mbx = __builtin_amdgcn_readfirstlane(mbx), mby = __builtin_amdgcn_readfirstlane(mby); // (wave-uniform: say so, or every address below is per-ln arithmetic)
        int ln = lane;
        asm volatile("" : "+v"(ln)); // (keeps lane-dependent addresses of the body from being hoisted out of the loop and spilled)
        // one macroblock: wait for the intra macroblocks it predicts from, reconstruct, publish
        const int c = mbx >> 6, k = mbx & 63;
        if (ln < 32) reinterpret_cast<uint32_t *>(&ws->rec)[ln] = reinterpret_cast<const uint32_t *>(recs + static_cast<uint64_t>(mby) * wmb + mbx)[ln];
        WAVE_SYNC();
        if (sparse && mbx > 0) { // (rows dealt out macroblock by macroblock: the left neighbour may be another wavefront's)
            const unsigned long long bit = 1ull << ((mbx - 1) & 63);
            while (__hip_atomic_load(&sh.pend[mby][(mbx - 1) >> 6], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & bit) __builtin_amdgcn_s_sleep(1);
        }
        if (mby > 0) { // the intra macroblocks among (mbx-1 .. mbx+1, mby-1) must be done
            const int xl = max(mbx - 1, 0), xr = min(mbx + 1, wmb - 1);
            const int c0 = xl >> 6, c1 = xr >> 6;
            const unsigned long long span = ((xr - xl + 1) >= 64 ? ~0ull : ((1ull << (xr - xl + 1)) - 1));
            const unsigned long long m0 = span << (xl & 63), m1 = c1 != c0 ? span >> (64 - (xl & 63)) : 0ull;
            while ((__hip_atomic_load(&sh.pend[mby - 1][c0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & m0) ||
                   (m1 && (__hip_atomic_load(&sh.pend[mby - 1][c1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & m1)))
                __builtin_amdgcn_s_sleep(1);
        }
        { // coefficient blocks of this macroblock: pool -> dense LDS layout (absent blocks are zero)
            const uint32_t cmask = ws->rec.coef_mask;
            uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
            if (ln < MI_COEF_BLOCKS && ((cmask >> ln) & 1)) {
                const uint4 *src = reinterpret_cast<const uint4 *>(coefs) + 2 * (static_cast<size_t>(ws->rec.coef_off) + __builtin_popcount(cmask & ((1u << ln) - 1u)));
                c0 = src[0], c1 = src[1];
            }
            if (ln < MI_COEF_BLOCKS) reinterpret_cast<uint4 *>(ws->coef)[2 * ln] = c0, reinterpret_cast<uint4 *>(ws->coef)[2 * ln + 1] = c1;
            WAVE_SYNC();
        }
        intra_mb<false>(ln, ws, &ws->rec, ws->coef, &sh.sc, py, pcb, pcr, W, mbx, mby);
        // done: the release orders this wavefront's sample stores before the bit is cleared
        if (ln == 0) __hip_atomic_fetch_and(&sh.pend[mby][c], ~(1ull << k), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// K3 spread over several workgroups per picture, for launches with fewer pictures than the chip has CUs: workgroup = a band of
// consecutive macroblock rows, wavefront w of it owns rows r0 + w, r0 + w + nwaves, ...  Inside a band the LDS bit masks order
// the rows as in k_intra.  The first row of a band waits for the intra macroblocks of the row above -- another workgroup's --
// through one agent-scope flag word per macroblock column (tag = this launch's epoch), which the band above sets after it has
// drained the macroblock's write-through sample stores.  A band waits only for the band above it and bands draw their
// (picture, band) from a ticket counter in that order, so the workgroup being waited for is always running.
extern "C" __global__ void __launch_bounds__(MI_INTRA_WAVES * 64) k_intra_x(const uint32_t *pic_list, const PicDesc *pics, const FramePool *pools, const DevTables *tab,
                                                                            const MbRec *mbrec, const int16_t *coefs, uint32_t *xdone_, uint32_t epoch, int nbands,
                                                                            uint32_t *ticket, uint32_t ticket_base, int wmb_max, uint32_t *xstatus, int wpr,
                                                                            const unsigned long long *intramask) {
    // wpr ("wavefronts per row"): the intra macroblocks of a row are dealt round-robin to wpr wavefronts.  In P / B pictures they are
    // few and mostly independent of each other, so the busiest row -- which bounds the kernel -- finishes wpr times sooner; a
    // macroblock whose left neighbour is an intra one too waits for that neighbour's bit like it waits for the row above.
    __shared__ IntraShared sh;
    __shared__ unsigned long long s_intra[MI_INTRA_MAX_ROWS][MI_INTRA_MAX_CHUNKS]; // sh.pend as pass 1 left it: which macroblocks K3 owns
    __shared__ uint32_t s_ticket;
    const int tid = static_cast<int>(threadIdx.x), lane = tid & 63, wave = tid >> 6, nthreads = static_cast<int>(blockDim.x), nwaves = nthreads >> 6;
    if (tid == 0) s_ticket = atomicAdd(ticket, 1u) - ticket_base;
    __syncthreads();
    const uint32_t tk = s_ticket, pic_i = tk / static_cast<uint32_t>(nbands);
    const int band = static_cast<int>(tk - pic_i * static_cast<uint32_t>(nbands));
    const PicDesc *pd = &pics[pic_list[pic_i]];
    const int wmb = static_cast<int>(pd->wmb), hmb = static_cast<int>(pd->hmb);
    const int r0 = band * hmb / nbands, r1 = (band + 1) * hmb / nbands;
    if (r0 >= r1) return; // (pictures smaller than the launch's largest can leave bands empty)
    const int W = static_cast<int>(pd->pitch); // (see k_intra)
    const uint32_t par_off = pd->field == 2 ? pd->pitch >> 1 : 0u;
    g8 *py = (g8 *)(pd->pool_base + static_cast<uint64_t>(pd->slot) * pd->slot_bytes) + par_off;
    g8 *pcb = py - par_off + pd->plane + (par_off >> 1), *pcr = pcb + (pd->plane >> 2);
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&tab->scaling[pd->scaling_set]);
        for (int i = tid; i < static_cast<int>(sizeof(ScalingSet) / 4); i += nthreads) reinterpret_cast<uint32_t *>(&sh.sc)[i] = src[i];
    }
    const int nchunks = (wmb + 63) >> 6;
    const MbRec *recs = mbrec + pd->mb_base;
    typedef __attribute__((address_space(1))) uint32_t gflag;
    int pband = band - 1; // the band that owns row r0 - 1
    while (pband > 0 && pband * hmb / nbands == (pband + 1) * hmb / nbands) pband--;
    gflag *const xin = (gflag *)xdone_ + (static_cast<size_t>(pic_i) * nbands + (pband > 0 ? pband : 0)) * static_cast<size_t>(wmb_max);
    gflag *const xout = (gflag *)xdone_ + (static_cast<size_t>(pic_i) * nbands + band) * static_cast<size_t>(wmb_max);
    // ---- pass 1: intra masks of the band's rows and of the row above it (that one is never cleared here: it says which flags to wait for) ----
    {
        const int rf = r0 > 0 ? r0 - 1 : 0;
        for (int i = tid; i < (r1 - rf) * nchunks; i += nthreads) {
            const int mby = rf + i / nchunks, c = i % nchunks;
            const unsigned long long m = intra_bits(intramask, pd->mb_base + static_cast<unsigned long long>(mby) * wmb + c * 64, wmb - c * 64);
            sh.pend[mby][c] = m, s_intra[mby][c] = m;
        }
    }
    __syncthreads();
    IntraWave *ws = &sh.w[wave];
    const int row_slot = wave / wpr, turn = wave - row_slot * wpr, row_slots = nwaves / wpr;
    for (int mby = r0 + row_slot; mby < r1; mby += row_slots) {
        const MbRec *row = recs + static_cast<uint64_t>(mby) * wmb;
        const bool publish = mby == r1 - 1 && r1 < hmb; // the band below waits for this row
        int nth = 0; // ordinal of the intra macroblock inside the row
        for (int c = 0; c < nchunks; c++) {
            const unsigned long long row_mask = s_intra[mby][c]; // (sh.pend loses bits as the row's other wavefronts finish macroblocks)
            unsigned long long mask = row_mask;
            while (mask) {
                const int k = __ffsll(static_cast<long long>(mask)) - 1;
                mask &= mask - 1;
                if (wpr > 1 && (nth++ % wpr) != turn) continue; // another wavefront's macroblock
                const int mbx = c * 64 + k;
                int ln = lane;
                asm volatile("" : "+v"(ln)); // (keeps lane-dependent addresses of the body from being hoisted out of the loops and spilled)
                if (ln < 32) reinterpret_cast<uint32_t *>(&ws->rec)[ln] = reinterpret_cast<const uint32_t *>(row + mbx)[ln];
                WAVE_SYNC();
                if (wpr > 1 && mbx > 0) { // the left neighbour, if it is an intra macroblock, belongs to another wavefront of this row
                    const int xl = mbx - 1;
                    const unsigned long long bit = 1ull << (xl & 63);
                    if (s_intra[mby][xl >> 6] & bit)
                        while (__hip_atomic_load(&sh.pend[mby][xl >> 6], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & bit) __builtin_amdgcn_s_sleep(1);
                }
                if (mby > 0) {
                    const int xl = max(mbx - 1, 0), xr = min(mbx + 1, wmb - 1);
                    const int c0 = xl >> 6, c1 = xr >> 6;
                    const unsigned long long span = ((xr - xl + 1) >= 64 ? ~0ull : ((1ull << (xr - xl + 1)) - 1));
                    const unsigned long long m0 = span << (xl & 63), m1 = c1 != c0 ? span >> (64 - (xl & 63)) : 0ull;
                    if (mby > r0) {
                        while ((__hip_atomic_load(&sh.pend[mby - 1][c0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & m0) ||
                               (m1 && (__hip_atomic_load(&sh.pend[mby - 1][c1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & m1)))
                            __builtin_amdgcn_s_sleep(1);
                    } else { // another workgroup's row: ln i watches the flag of column xl + i if that macroblock is an intra one
                        const int col = xl + ln;
                        const bool need = ln < 3 && col <= xr && ((sh.pend[mby - 1][col >> 6] >> (col & 63)) & 1ull);
                        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                        for (;;) {
                            const uint32_t v = need ? __hip_atomic_load(xin + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
                            if (__builtin_amdgcn_ballot_w64(v != epoch) == 0) break;
                            __builtin_amdgcn_s_sleep(2);
                            if (__builtin_amdgcn_s_memrealtime() - t_start > 400000000ull) { // 4 s: report instead of hanging the GPU
                                if (ln == 0) atomicExch(xstatus, 0x3D000000u | static_cast<uint32_t>(mby));
                                break;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (no instruction: the sample loads below stay below the poll)
                    }
                }
                {
                    const uint32_t cmask = ws->rec.coef_mask;
                    uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
                    if (ln < MI_COEF_BLOCKS && ((cmask >> ln) & 1)) {
                        const uint4 *src = reinterpret_cast<const uint4 *>(coefs) + 2 * (static_cast<size_t>(ws->rec.coef_off) + __builtin_popcount(cmask & ((1u << ln) - 1u)));
                        c0 = src[0], c1 = src[1];
                    }
                    if (ln < MI_COEF_BLOCKS) reinterpret_cast<uint4 *>(ws->coef)[2 * ln] = c0, reinterpret_cast<uint4 *>(ws->coef)[2 * ln + 1] = c1;
                    WAVE_SYNC();
                }
                intra_mb<true>(ln, ws, &ws->rec, ws->coef, &sh.sc, py, pcb, pcr, W, mbx, mby);
                if (publish) { // every sample store of this wavefront has left the CU before the flag does
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (ln == 0) __hip_atomic_store(xout + mbx, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (ln == 0) __hip_atomic_fetch_and(&sh.pend[mby][c], ~(1ull << k), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

// ================================================================== K6: crop + pack to tight I420
// One launch packs any number of frames (a whole batch): blockIdx.x = frame, blockIdx.y = a chunk of its rows (luma rows,
// then the Cb rows, then the Cr rows).  A thread moves 16 bytes when source and destination rows are 16-byte aligned
// (1080p: always), single bytes otherwise.
extern "C" __global__ void __launch_bounds__(256) k_pack(const PackDesc *descs, uint8_t *dst, int rows_per_block) {
    const PackDesc pd = descs[blockIdx.x];
    const int w = static_cast<int>(pd.w), h = static_cast<int>(pd.h), W = static_cast<int>(pd.W), H = static_cast<int>(pd.H);
    const int nrows = 2 * h; // h luma rows + h/2 + h/2 chroma rows
    const uint8_t *src = reinterpret_cast<const uint8_t *>(pd.src);
    uint8_t *out = dst + pd.dst_off;
    const int r0 = static_cast<int>(blockIdx.y) * rows_per_block, r1 = min(r0 + rows_per_block, nrows);
    for (int r = r0; r < r1; r++) {
        const uint8_t *s;
        uint8_t *d;
        int n;
        if (r < h) {
            s = src + static_cast<size_t>(r + pd.y0) * W + pd.x0, d = out + static_cast<size_t>(r) * w, n = w;
        } else {
            const int c = r - h >= h / 2, rc = r - h - c * (h / 2);
            s = src + static_cast<size_t>(W) * H + static_cast<size_t>(c) * (W / 2) * (H / 2) + static_cast<size_t>(rc + pd.y0 / 2) * (W / 2) + pd.x0 / 2;
            d = out + static_cast<size_t>(w) * h + static_cast<size_t>(c) * (w / 2) * (h / 2) + static_cast<size_t>(rc) * (w / 2), n = w / 2;
        }
        if ((((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) && (n & 15) == 0) {
            for (int i = static_cast<int>(threadIdx.x) * 16; i < n; i += 256 * 16) *reinterpret_cast<uint4 *>(d + i) = *reinterpret_cast<const uint4 *>(s + i);
        } else {
            for (int i = static_cast<int>(threadIdx.x); i < n; i += 256) d[i] = s[i];
        }
    }
}

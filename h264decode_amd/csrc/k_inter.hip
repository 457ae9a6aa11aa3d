// h264decode_amd/csrc/k_inter.hip -- K4: inter prediction + residual of P and B macroblocks (ITU-T H.264 8.4.2, 8.5), gfx950.
//
// Work split: ONE LANE PER 4x4 BLOCK.  A wavefront takes four consecutive macroblocks (16 lanes each); lane b of a
// macroblock owns luma block b (raster) -- its 16 samples as four packed dwords -- and the 2x2 chroma samples under it.
// 4x4 is the granularity of H.264 motion (one vector per 4x4 block in the record), so there is no partition logic and no
// "uniform motion" special case: every lane fetches the window its own vector needs straight from the reference picture
// (unaligned dword loads, rows and columns clamped to the picture as 8.4.2.2 prescribes) and nothing is computed twice.
//
// Arithmetic is packed-byte / packed-16 wherever the sample precision allows:
//   horizontal 6-tap (1,-5,20,20,-5,1): two v_dot4_i32_i8 per sample on the window bytes (biased by 128: the taps sum to 32),
//                                        the sliding window is v_alignbyte_b32;
//   vertical 6-tap:                     v_pk_add_u16 / v_pk_mad_u16 on two columns at a time (|sum| <= 10710 fits 16 bits);
//   centre position j:                  32-bit on the unrounded horizontal sums (they do not fit 16 bits after the second pass);
//   quarter positions, default bi-pred: v_lerp_u8 -- (a + b + 1) >> 1 on four samples at once;
//   chroma 1/8 bilinear:                v_perm_b32 gathers (A,B,C,D), one v_dot4_u32_u8 with the four weights.
// Residual: a lane dequantises and inverse-transforms its own 4x4 block entirely in registers (no LDS pass); 8x8 transforms and
// the chroma 4x4 blocks (8 of them under 16 lanes) exchange through a small LDS tile.
//
// The reference has none of this (README.md:10 "Macroblock to YCbCr image decoding" is a TODO).
#include <hip/hip_runtime.h>
#include <cstddef>
#include "mi_kernels.h"

#define WAVE_SYNC()                                            \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

typedef __attribute__((address_space(1))) uint8_t g8;
typedef __attribute__((address_space(1))) uint16_t g16;
typedef __attribute__((address_space(1))) uint32_t g32;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef short s2 __attribute__((ext_vector_type(2)));

// The clamped value is made opaque on purpose: left alone, the compiler fuses "shift, clamp, pack two bytes" into gfx950's
// v_ashr_pk_u8_i32 and then ORs further bytes into the upper half of its result -- but the instruction leaves whatever the
// destination register held there (measured: tools/ubench/, stale window bytes ended up in samples 2 and 3 of every row).
__device__ __forceinline__ int clip255(int v) {
    v = min(max(v, 0), 255);
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ uint32_t lerp8(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); } // (a + b + 1) >> 1 per byte
__device__ __forceinline__ uint32_t align8(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
    return static_cast<uint32_t>(a) | (static_cast<uint32_t>(b) << 8) | (static_cast<uint32_t>(c) << 16) | (static_cast<uint32_t>(d) << 24);
}
__device__ __forceinline__ s2 as_s2(uint32_t v) { return __builtin_bit_cast(s2, v); }
__device__ __forceinline__ uint32_t as_u32(s2 v) { return __builtin_bit_cast(uint32_t, v); }

// Clip1(prediction + residual) of four packed samples, two at a time in 16-bit halves: the residuals are narrowed with saturation (v_cvt_pk_i16_i32: a residual
// beyond +-32767 ends at 0 or 255 either way; tools/ubench/intrin_probe.hip shows the instruction saturating), the sums cannot leave 16 bits.
__device__ __forceinline__ uint32_t add_clip4(uint32_t p, int r0, int r1, int r2, int r3) {
    const s2 z = {0, 0}, m = {255, 255};
    const s2 lo = __builtin_elementwise_add_sat(as_s2(__builtin_amdgcn_perm(0u, p, 0x0c010c00u)), __builtin_bit_cast(s2, __builtin_amdgcn_cvt_pk_i16(r0, r1)));
    const s2 hi = __builtin_elementwise_add_sat(as_s2(__builtin_amdgcn_perm(0u, p, 0x0c030c02u)), __builtin_bit_cast(s2, __builtin_amdgcn_cvt_pk_i16(r2, r3)));
    return __builtin_amdgcn_perm(as_u32(__builtin_elementwise_min(__builtin_elementwise_max(hi, z), m)), as_u32(__builtin_elementwise_min(__builtin_elementwise_max(lo, z), m)), 0x06040200u);
}

// ------------------------------------------------------------------ 1-D inverse transforms / scaling (8.5.12, 8.5.13)
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
// The same with the case distinction taken out of the per-coefficient work: ((c * ls + rnd) >> sr) << sl with (sl, sr, rnd) = (per - 4, 0, 0) or
// (0, 4 - per, 1 << (3 - per)) -- one multiply-add and two shifts by a per-lane amount, no select (one of the shifts is by 0).
struct Scale4 {
    int sl, sr, rnd;
    __device__ __forceinline__ explicit Scale4(int qp) {
        const int per = qp / 6;
        sl = max(per - 4, 0), sr = max(4 - per, 0), rnd = (1 << sr) >> 1;
    }
    __device__ __forceinline__ int operator()(int c, int ls) const { return ((c * ls + rnd) >> sr) << sl; }
};
__device__ __forceinline__ int scale8(int c, int ls, int qp) { // 8.5.13
    int per = qp / 6;
    return per >= 6 ? (c * ls) << (per - 6) : (c * ls + (1 << (5 - per))) >> (6 - per);
}
// the 16 coefficients of a 32-byte block: two 16-byte loads, sign-extended halves
__device__ __forceinline__ void load_block(const int16_t *coefs, uint32_t blk, int (&c)[16]) {
    const v4u *src = reinterpret_cast<const v4u *>(coefs) + 2 * static_cast<size_t>(blk);
    const v4u a = src[0], b = src[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; i++) c[2 * i] = static_cast<int>(static_cast<int16_t>(w[i] & 0xFFFFu)), c[2 * i + 1] = static_cast<int>(w[i]) >> 16;
}

struct InterLds { // per macroblock of the wavefront
    int32_t t8[4][64];    // 8x8 transform: row-pass output of the four 8x8 blocks, then their residual
    int16_t cres[2][64];  // chroma residual, raster 8x8 per plane
};

// One list's prediction of this lane's block: P[r] = luma row r (4 samples), C[c] = plane c: row 0 in the low half, row 1 in the high half.
// base: the stream's frame pool; loff / coff: byte offset of the reference picture's first luma / Cb row (a field: the first row of its
// parity); cr_delta: from a Cb sample to the Cr sample at the same place; pitch: bytes from one luma row of the reference picture to the next
// (chroma: half); (px, py): the block's luma position; W x H: the picture; mvy_c: the vertical vector chroma uses (8.4.1.4: in field pictures it
// differs from the luma vector by a quarter sample when the reference field has the other parity).
__device__ __forceinline__ void predict(const g8 *base, uint32_t loff, uint32_t coff, uint32_t cr_delta, uint32_t pitch, int px, int py, int mvx, int mvy, int mvy_c, int W, int H,
                                        uint32_t (&P)[4], uint32_t (&C)[2]) {
    const g8 *ref = base + loff;
    // ---------------- luma (8.4.2.2.1): window slot j = row y0 - 2 + j, byte k = column x0 - 2 + k
    {
        const int fx = mvx & 3, fy = mvy & 3, x0 = px + (mvx >> 2), y0 = py + (mvy >> 2);
        const bool needH = fx != 0, needV = fy != 0;
        const int xs = x0 - 2;
        // columns: a 12-byte span that lies inside the row; where the window sticks out, its bytes are picked from the span by
        // v_perm with the clamped positions (xInt = Clip3(0, W - 1, ...)): same loads, three more instructions per row
        const int xb = min(max(xs, 0), W - 12);
        const bool fix = xb != xs;
        uint32_t selw[3] = {0x03020100u, 0x03020100u, 0x03020100u};
        int srcw[3] = {0, 1, 2};
        if (__builtin_amdgcn_ballot_w64(fix) != 0) {
#pragma unroll
            for (int m = 0; m < 3; m++) {
                int idx[4];
#pragma unroll
                for (int k = 0; k < 4; k++) idx[k] = min(max(xs + 4 * m + k, 0), W - 1) - xb; // 0..11, non-decreasing
                const int s0 = idx[0] >> 2;                                                    // the two span dwords this output dword draws from: s0, s0 + 1
                srcw[m] = s0;
                selw[m] = static_cast<uint32_t>(idx[0] - 4 * s0) | (static_cast<uint32_t>(idx[1] - 4 * s0) << 8) | (static_cast<uint32_t>(idx[2] - 4 * s0) << 16) |
                          (static_cast<uint32_t>(idx[3] - 4 * s0) << 24);
            }
        }
        uint32_t d0[9], d1[9], d2[9];
#pragma unroll
        for (int j = 0; j < 9; j++) {
            d0[j] = d1[j] = d2[j] = 0;
            const bool row_needed = (j >= 2 && j <= 5) || needV;
            if (row_needed) {
                const int y = min(max(y0 - 2 + j, 0), H - 1);
                // (Round 4 measured what else could feed the window, profiles/r04_k4_k5_counters.txt -- wavefronts parked at s_waitcnt 61 % of their cycles, texture-
                // address units 67 % busy --: one unaligned 16- / 8-byte load per row 1.09 -> 1.15 ms (fractional motion 1.13 -> 1.31); one dword-ALIGNED 16- / 12-byte
                // load + v_alignbyte 1.30 / 1.47 ms; the macroblock's 21 x 21 window staged once through LDS for macroblocks that move as one (0.7 KB instead of 2.1 KB
                // through the address path) 1.26 / 1.41 ms, bit-exact.  All slower: the kernel is bound by the LATENCY of its chain descriptor -> record -> samples at
                // five wavefronts per SIMD, and every one of them adds bytes or a stage to that chain; three dword loads per row in flight at once is the shortest.)
                const g8 *p = ref + (static_cast<uint32_t>(y) * pitch + static_cast<uint32_t>(xb));
                d0[j] = *reinterpret_cast<const g32 *>(p), d1[j] = *reinterpret_cast<const g32 *>(p + 4);
                if (needH || fix) d2[j] = *reinterpret_cast<const g32 *>(p + 8);
            }
        }
        if (__builtin_amdgcn_ballot_w64(fix) != 0) {
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const uint32_t a = d0[j], b = d1[j], c = d2[j];
                uint32_t o[3];
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    const uint32_t lo = srcw[m] == 0 ? a : (srcw[m] == 1 ? b : c), hi = srcw[m] == 0 ? b : c; // (s0 == 2: every index is inside dword 2)
                    o[m] = __builtin_amdgcn_perm(hi, lo, selw[m]);
                }
                d0[j] = fix ? o[0] : a, d1[j] = fix ? o[1] : b, d2[j] = fix ? o[2] : c;
            }
        }
        const bool isJ = (fx == 2 && fy != 0) || (fy == 2 && fx != 0); // j and its four quarter neighbours
        const uint32_t sv = fy == 3 ? 1u : 0u;                          // integer / half-sample row the quarter positions lean on: slot 2 or 3
        const uint32_t cs = fx == 3 ? 3u : 2u;                          // ... and column: byte 2 or 3
        uint32_t X[4], Y[4];
        // G: integer samples at (column byte 2, slots 2..5)
#pragma unroll
        for (int r = 0; r < 4; r++) X[r] = Y[r] = align8(d1[2 + r], d0[2 + r], 2);
        // ---- horizontal half samples b (rows slots 2 + sv ..): every position with fx != 0 that is not in the j family
        const bool useH = needH && !isJ;
        if (__builtin_amdgcn_ballot_w64(useH) != 0) {
            uint32_t Hh[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t a = (sv ? d0[3 + r] : d0[2 + r]) ^ 0x80808080u, b = (sv ? d1[3 + r] : d1[2 + r]) ^ 0x80808080u, c = (sv ? d2[3 + r] : d2[2 + r]) ^ 0x80808080u;
                int h[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t A = i ? align8(b, a, i) : a, Bq = i ? align8(c, b, i) : b; // window bytes i..i+3 and i+4..i+7
                    h[i] = __builtin_amdgcn_sdot4(static_cast<int>(A), 0x1414FB01, __builtin_amdgcn_sdot4(static_cast<int>(Bq), 0x000001FB, 4096 + 16, false), false);
                    h[i] = clip255(h[i] >> 5);
                }
                Hh[r] = pack4(h[0], h[1], h[2], h[3]);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (useH) X[r] = Hh[r];
                if (useH && fy == 0 && fx == 2) Y[r] = Hh[r];
                if (useH && fy == 0 && fx != 2) Y[r] = align8(d1[2 + r], d0[2 + r], cs); // quarter: lean on G (column 2) or H (column 3)
            }
        }
        // ---- vertical half samples h (column byte cs): fx == 0, the diagonal quarters, and the j neighbours (1,2) / (3,2)
        const bool useV = needV && (!isJ || (fy == 2 && fx != 2));
        if (__builtin_amdgcn_ballot_w64(useV) != 0) {
            s2 lo[9], hi[9]; // the four columns of every slot as 16-bit pairs
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const uint32_t w = cs == 3 ? align8(d1[j], d0[j], 3) : align8(d1[j], d0[j], 2);
                lo[j] = as_s2(__builtin_amdgcn_perm(0u, w, 0x0c010c00u)), hi[j] = as_s2(__builtin_amdgcn_perm(0u, w, 0x0c030c02u));
            }
            uint32_t Vh[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const s2 vl = (lo[r] + lo[r + 5]) - (lo[r + 1] + lo[r + 4]) * static_cast<short>(5) + (lo[r + 2] + lo[r + 3]) * static_cast<short>(20);
                const s2 vh = (hi[r] + hi[r + 5]) - (hi[r + 1] + hi[r + 4]) * static_cast<short>(5) + (hi[r + 2] + hi[r + 3]) * static_cast<short>(20);
                const s2 z = {0, 0}, m = {255, 255};
                const s2 cl = __builtin_elementwise_min(__builtin_elementwise_max((vl + static_cast<short>(16)) >> 5, z), m);
                const s2 ch = __builtin_elementwise_min(__builtin_elementwise_max((vh + static_cast<short>(16)) >> 5, z), m);
                Vh[r] = __builtin_amdgcn_perm(as_u32(ch), as_u32(cl), 0x06040200u);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (useV && !needH) X[r] = Vh[r];
                if (useV && !needH && fy != 2) Y[r] = sv ? align8(d1[3 + r], d0[3 + r], 2) : align8(d1[2 + r], d0[2 + r], 2); // quarter: lean on G (slot 2) or the row below
                if (useV && (needH || fy == 2)) Y[r] = Vh[r]; // (0,2): both; diagonal quarters: b/s with h/m; (1,2) / (3,2): j with h/m
            }
        }
        // ---- the j family: unrounded horizontal sums of all nine slots, then the vertical taps on them
        if (__builtin_amdgcn_ballot_w64(isJ) != 0) {
            int hs[9][4];
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const uint32_t a = d0[j] ^ 0x80808080u, b = d1[j] ^ 0x80808080u, c = d2[j] ^ 0x80808080u;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t A = i ? align8(b, a, i) : a, Bq = i ? align8(c, b, i) : b;
                    hs[j][i] = __builtin_amdgcn_sdot4(static_cast<int>(A), 0x1414FB01, __builtin_amdgcn_sdot4(static_cast<int>(Bq), 0x000001FB, 4096, false), false);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int jv[4], bs[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int v = (hs[r][i] + hs[r + 5][i]) - 5 * (hs[r + 1][i] + hs[r + 4][i]) + 20 * (hs[r + 2][i] + hs[r + 3][i]);
                    jv[i] = clip255((v + 512) >> 10);
                    bs[i] = clip255(((sv ? hs[r + 3][i] : hs[r + 2][i]) + 16) >> 5); // b (slot 2) or s (slot 3) of this row
                }
                const uint32_t J = pack4(jv[0], jv[1], jv[2], jv[3]);
                if (isJ) X[r] = J;
                if (isJ && fx == 2 && fy == 2) Y[r] = J;
                if (isJ && fx == 2 && fy != 2) Y[r] = pack4(bs[0], bs[1], bs[2], bs[3]);
                // (fy == 2, fx odd: Y is the vertical half sample, set above)
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) P[r] = lerp8(X[r], Y[r]);
    }
    // ---------------- chroma (8.4.2.2.2): 2x2 samples per plane, 1/8 bilinear
    {
        const int Wc = W >> 1, Hc = H >> 1;
        const int xf = mvx & 7, yf = mvy_c & 7, cx = (px >> 1) + (mvx >> 3), cy = (py >> 1) + (mvy_c >> 3);
        const int xb = min(max(cx, 0), Wc - 4);
        uint32_t sel = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) sel |= static_cast<uint32_t>(min(max(cx + k, 0), Wc - 1) - xb) << (8 * k); // window byte k <- span byte
        const uint32_t wts = static_cast<uint32_t>((8 - xf) * (8 - yf)) | (static_cast<uint32_t>(xf * (8 - yf)) << 8) | (static_cast<uint32_t>((8 - xf) * yf) << 16) |
                             (static_cast<uint32_t>(xf * yf) << 24);
        const uint32_t cpitch = pitch >> 1;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const g8 *cp = base + coff + (c ? cr_delta : 0u);
            uint32_t w[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int y = min(max(cy + j, 0), Hc - 1);
                w[j] = __builtin_amdgcn_perm(0u, *reinterpret_cast<const g32 *>(cp + (static_cast<uint32_t>(y) * cpitch + static_cast<uint32_t>(xb))), sel);
            }
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 2; i++) // (A, B, C, D) = (w[j][i], w[j][i+1], w[j+1][i], w[j+1][i+1])
                    o[2 * j + i] = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(w[j + 1], w[j], i ? 0x06050201u : 0x05040100u), wts, 32u, false) >> 6;
            C[c] = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
        }
    }
}

// 8.4.2.3 on four packed samples: a = list-0 prediction, b = list-1 prediction; mode 1 explicit, 2 implicit (0 is handled by the caller)
__device__ __forceinline__ uint32_t weigh4(uint32_t a, uint32_t b, bool u0, bool u1, int wmode, int ld, int w0, int o0, int w1, int o1, int iw1) {
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int pa = static_cast<int>((a >> (8 * i)) & 255u), pb = static_cast<int>((b >> (8 * i)) & 255u);
        int v;
        if (u0 && u1) {
            if (wmode == 1)
                v = clip255(((pa * w0 + pb * w1 + (1 << ld)) >> (ld + 1)) + ((o0 + o1 + 1) >> 1));
            else if (wmode == 2)
                v = clip255((pa * (64 - iw1) + pb * iw1 + 32) >> 6);
            else
                v = (pa + pb + 1) >> 1;
        } else {
            v = u0 ? pa : pb;
            const int w = u0 ? w0 : w1, o = u0 ? o0 : o1;
            if (wmode == 1) v = ld >= 1 ? clip255(((v * w + (1 << (ld - 1))) >> ld) + o) : clip255(v * w + o);
        }
        out |= static_cast<uint32_t>(v) << (8 * i);
    }
    return out;
}

template <bool B>
__device__ __forceinline__ void inter4(InterLds *lds, const uint32_t *pic_list, const PicDesc *pics, const SliceDesc *slices, const DevTables *tab, const MbRec *mbrec,
                                       const int16_t *coefs, int groups_per_pic_log2, int n_blocks, const BSliceExt *bexts, const MbMv1 *mbmv1) {
    const int lane = static_cast<int>(threadIdx.x) & 63, sub = lane >> 4, b = lane & 15, bx = b & 3, by = b >> 2, q8 = ((by >> 1) << 1) | (bx >> 1);
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2; an XCD gets a contiguous run of
    // macroblock groups (whole pictures), so that the reference rows neighbouring blocks share come from the same L2
    const uint32_t per_xcd = gridDim.x >> 3, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lb = ((blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * MI_K4_WG + wave;
    lds += 4 * wave;
    if (lb >= static_cast<uint32_t>(n_blocks)) return;
    const PicDesc *pd = &pics[pic_list[lb >> groups_per_pic_log2]];
    const int wmb = static_cast<int>(pd->wmb), hmb = static_cast<int>(pd->hmb);
    const int mb = static_cast<int>((lb & ((1u << groups_per_pic_log2) - 1u)) * 4u) + sub;
    const bool valid = mb < wmb * hmb;
    const uint64_t mbi = pd->mb_base + (valid ? mb : 0);
    const uint32_t *rw = reinterpret_cast<const uint32_t *>(mbrec + mbi); // the record as dwords (mi_types.h: MbRec)
    const uint32_t h0 = rw[0];
    const int type = static_cast<int>(h0 & 255u);
    const bool inter = valid && MB_IS_INTER(type);
    if (__builtin_amdgcn_ballot_w64(inter) == 0) return;
    const uint32_t h1 = rw[1];
    const int t8x8 = static_cast<int>((h0 >> 8) & 255u), qp = static_cast<int>((h0 >> 16) & 255u), cbp = static_cast<int>((h1 >> 8) & 255u);
    const uint32_t cmask = inter ? rw[29] : 0u, coef_off = rw[28];
    const int W = wmb * 16, H = hmb * 16;
    const int mby = static_cast<int>(__umulhi(static_cast<uint32_t>(valid ? mb : 0), pd->inv_wmb)), mbx = (valid ? mb : 0) - mby * wmb;
    const int px = mbx * 16 + bx * 4, py = mby * 16 + by * 4;
    const g8 *pool = (const g8 *)pd->pool_base;
    const uint32_t slot_bytes = static_cast<uint32_t>(pd->slot_bytes);
    const int max_slot = static_cast<int>(pd->n_slots) - 1;
    // the picture's place in its frame slot (PicDesc): a field picture lives in the rows of its parity, and its references are fields
    const uint32_t pitch = pd->pitch, plane = pd->plane, cr_delta = plane >> 2;
    const int fld = static_cast<int>(pd->field);
    const uint32_t fpitch = fld ? pitch >> 1 : 0u; // bytes from a frame row to the next = where the bottom field starts (0: frame picture, no parities)
    // byte offsets of a reference picture's first luma / Cb row, and the chroma vector offset of Table 8-9 / 8-10 (reference field of the other parity)
    auto ref_place = [&](int rs, uint32_t &lo, uint32_t &co, int &cadj) {
        const uint32_t par = fld ? static_cast<uint32_t>(rs >> 14) & 1u : 0u;
        const uint32_t so = static_cast<uint32_t>(min(max(fld ? MI_REF_SLOT(rs) : rs, 0), max_slot)) * slot_bytes;
        lo = so + par * fpitch, co = so + plane + par * (fpitch >> 1);
        cadj = fld ? (static_cast<int>(par) - (fld - 1)) * -2 : 0; // bottom field from a top field: +2; top from bottom: -2
    };
    // this block's vectors / reference frames
#if defined(K4_EXP) && (K4_EXP & 4)
    const uint32_t mvw0 = rw[12 + b] & 0xFFFCFFFCu;
#else
    const uint32_t mvw0 = rw[12 + b];
#endif
    const int s0 = static_cast<int>(static_cast<int16_t>(reinterpret_cast<const uint16_t *>(rw)[18 + q8]));   // refslot[q8] (byte 36)
    int s1 = -1;
    uint32_t mvw1 = 0;
    if (B) {
        s1 = static_cast<int>(static_cast<int16_t>(reinterpret_cast<const uint16_t *>(rw)[60 + q8]));         // refslot1[q8] (byte 120)
        if (inter && s1 >= 0) mvw1 = reinterpret_cast<const uint32_t *>(mbmv1 + mbi)[b];
    }
    const bool u1 = B && s1 >= 0, u0 = s0 >= 0 || !u1; // a record without any usable reference is concealed from list 0
    uint32_t P[4] = {0, 0, 0, 0}, C[2] = {0, 0}, P1[4] = {0, 0, 0, 0}, C1[2] = {0, 0};
    if (inter && u0) {
        uint32_t lo, co;
        int cadj;
        ref_place(max(s0, 0), lo, co, cadj);
        const int mvy = static_cast<int>(mvw0) >> 16;
        predict(pool, lo, co, cr_delta, pitch, px, py, static_cast<int16_t>(mvw0 & 0xFFFFu), mvy, mvy + cadj, W, H, P, C);
    }
    if (B && __builtin_amdgcn_ballot_w64(inter && u1) != 0) {
        if (inter && u1) {
            uint32_t lo, co;
            int cadj;
            ref_place(s1, lo, co, cadj);
            const int mvy = static_cast<int>(mvw1) >> 16;
            predict(pool, lo, co, cr_delta, pitch, px, py, static_cast<int16_t>(mvw1 & 0xFFFFu), mvy, mvy + cadj, W, H, P1, C1);
        }
    }
    // ---- weighting (8.4.2.3) ----
    {
        const SliceDesc *sd = &slices[rw[11]];
        const BSliceExt *bx_ = (B && sd->slice_type == 1) ? &bexts[sd->bext] : nullptr;
        const int wmode = !inter ? 0 : (B ? (bx_ ? bx_->wp_mode : (sd->wp_flag ? 1 : 0)) : (pd->weighted_pred ? 1 : 0)); // 0 default, 1 explicit, 2 implicit
        const bool plain = wmode == 0 || (wmode == 2 && !(u0 && u1)); // implicit weights only act on bi-predicted blocks
        if (__builtin_amdgcn_ballot_w64(!plain) != 0) {
            int ref0 = static_cast<int>(static_cast<int8_t>(reinterpret_cast<const uint8_t *>(rw)[32 + q8])), ref1 = 0;
            if (B) ref1 = static_cast<int>(static_cast<int8_t>(reinterpret_cast<const uint8_t *>(rw)[16 + q8])); // ref_idx_l1 (MBREC_REF1)
            ref0 = max(ref0, 0) & (MI_MAX_REFS - 1), ref1 = max(ref1, 0) & (MI_MAX_REFS - 1);
            if (!plain) {
                int ld = 0, w0 = 1, o0 = 0, w1 = 1, o1 = 0, iw1 = 32, ldc = 0, cw0[2] = {1, 1}, co0[2] = {0, 0}, cw1[2] = {1, 1}, co1[2] = {0, 0};
                if (wmode == 1) {
                    ld = sd->luma_log2_denom, w0 = sd->wp_lw[ref0], o0 = sd->wp_lo[ref0], ldc = sd->chroma_log2_denom;
                    cw0[0] = sd->wp_cw[ref0][0], cw0[1] = sd->wp_cw[ref0][1], co0[0] = sd->wp_co[ref0][0], co0[1] = sd->wp_co[ref0][1];
                    if (bx_) {
                        w1 = bx_->wp_lw1[ref1], o1 = bx_->wp_lo1[ref1];
                        cw1[0] = bx_->wp_cw1[ref1][0], cw1[1] = bx_->wp_cw1[ref1][1], co1[0] = bx_->wp_co1[ref1][0], co1[1] = bx_->wp_co1[ref1][1];
                    }
                } else
                    iw1 = bx_->implicit_w1[ref0][ref1];
#pragma unroll
                for (int r = 0; r < 4; r++) P[r] = weigh4(P[r], P1[r], u0, u1, wmode, ld, w0, o0, w1, o1, iw1);
#pragma unroll
                for (int c = 0; c < 2; c++) C[c] = weigh4(C[c], C1[c], u0, u1, wmode, ldc, cw0[c], co0[c], cw1[c], co1[c], iw1);
            }
        }
        if (B && plain) { // default: the one prediction there is, or the rounded average of the two
#pragma unroll
            for (int r = 0; r < 4; r++) P[r] = u0 ? (u1 ? lerp8(P[r], P1[r]) : P[r]) : P1[r];
#pragma unroll
            for (int c = 0; c < 2; c++) C[c] = u0 ? (u1 ? lerp8(C[c], C1[c]) : C[c]) : C1[c];
        }
    }
    // ---- residual (8.5): this lane's 4x4 luma block in registers; 8x8 transforms and chroma blocks through LDS ----
    InterLds *ml = lds + sub;
    const ScalingSet *sc = &tab->scaling[pd->scaling_set];
    int res[16];
#pragma unroll
    for (int i = 0; i < 16; i++) res[i] = 0;
#if defined(K4_EXP) && (K4_EXP & 1) /* instruction-budget experiments (tools/k4_budget.sh; wrong pictures): 1 no luma residual, 2 no chroma residual, 4 integer vectors only */
    const bool has_l = false;
#else
    const bool has_l = inter && !t8x8 && ((cmask >> b) & 1u);
#endif
    if (__builtin_amdgcn_ballot_w64(has_l) != 0) {
        if (has_l) {
            int c[16];
            load_block(coefs, coef_off + __builtin_popcount(cmask & ((1u << b) - 1u)), c);
            const uint16_t *ls = sc->ls4[3][qp % 6]; // inter Y
            int t[16];
            const Scale4 sq(qp);
#pragma unroll
            for (int r = 0; r < 4; r++)
                inv4(sq(c[4 * r], ls[4 * r]), sq(c[4 * r + 1], ls[4 * r + 1]), sq(c[4 * r + 2], ls[4 * r + 2]), sq(c[4 * r + 3], ls[4 * r + 3]), t[4 * r], t[4 * r + 1], t[4 * r + 2],
                     t[4 * r + 3]);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int o0, o1, o2, o3;
                inv4(t[k], t[4 + k], t[8 + k], t[12 + k], o0, o1, o2, o3);
                res[k] = (o0 + 32) >> 6, res[4 + k] = (o1 + 32) >> 6, res[8 + k] = (o2 + 32) >> 6, res[12 + k] = (o3 + 32) >> 6;
            }
        }
    }
    // 8x8 transform (High profile): the four lanes of an 8x8 block take two of its rows each, then two of its columns
    const bool has_8 = inter && t8x8 && ((cbp >> q8) & 1);
    if (__builtin_amdgcn_ballot_w64(inter && t8x8) != 0) {
        const int j8 = ((by & 1) << 1) | (bx & 1); // this lane's place among the four: rows 2 j8, 2 j8 + 1; columns likewise
        if (has_8) {
            const uint16_t *ls = sc->ls8[1][qp % 6];
            const uint32_t blk = 4u * q8 + j8; // staging block: 16 coefficients = rows 2 j8, 2 j8 + 1 of the 8x8 block
            int c[16];
#pragma unroll
            for (int i = 0; i < 16; i++) c[i] = 0;
            if ((cmask >> blk) & 1u) load_block(coefs, coef_off + __builtin_popcount(cmask & ((1u << blk) - 1u)), c);
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int row = 2 * j8 + rr;
                int d[8], o[8];
#pragma unroll
                for (int k = 0; k < 8; k++) d[k] = scale8(c[8 * rr + k], ls[row * 8 + k], qp);
                inv8(d, o);
#pragma unroll
                for (int k = 0; k < 8; k++) ml->t8[q8][row * 8 + k] = o[k];
            }
        }
        WAVE_SYNC();
        if (has_8) {
            int colres[2][8];
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                const int col = 2 * j8 + cc;
                int d[8], o[8];
#pragma unroll
                for (int k = 0; k < 8; k++) d[k] = ml->t8[q8][k * 8 + col];
                inv8(d, o);
#pragma unroll
                for (int k = 0; k < 8; k++) colres[cc][k] = (o[k] + 32) >> 6;
            }
#pragma unroll
            for (int cc = 0; cc < 2; cc++)
#pragma unroll
                for (int k = 0; k < 8; k++) ml->t8[q8][k * 8 + 2 * j8 + cc] = colres[cc][k];
        }
        WAVE_SYNC();
        if (has_8) { // this lane's 4x4 quadrant of the 8x8 residual
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int k = 0; k < 4; k++) res[4 * r + k] = ml->t8[q8][((by & 1) * 4 + r) * 8 + (bx & 1) * 4 + k];
        }
    }
    // chroma: lanes 0..7 of a macroblock transform its eight 4x4 chroma blocks, every lane then picks up the 2x2 residual under it
    const int cbp_c = cbp >> 4;
#if defined(K4_EXP) && (K4_EXP & 2)
    const bool has_c = false;
#else
    const bool has_c = inter && cbp_c != 0;
#endif
    int cr[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (__builtin_amdgcn_ballot_w64(has_c) != 0 && __builtin_amdgcn_ballot_w64(has_c && (cbp_c & 2)) == 0) {
        // No macroblock of the wavefront has chroma AC coefficients (the rule: coded_block_pattern says 1 for most macroblocks with chroma residual at all).  A block
        // with its DC coefficient only transforms into sixteen times (dcC + 32) >> 6, so every lane works out the DC of the chroma block its 2x2 samples lie in, for
        // both planes: no 4x4 transform, no LDS exchange.
        const uint32_t db = MI_COEF_CDC / 16;
        if (has_c && ((cmask >> db) & 1u)) {
            const uint32_t *dcw = reinterpret_cast<const uint32_t *>(coefs) + 8 * static_cast<size_t>(coef_off + __builtin_popcount(cmask & ((1u << db) - 1u)));
            const v4u w = *reinterpret_cast<const v4u *>(dcw);
            const int blk = (by & 2) | (bx >> 1);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint32_t w0 = c ? w.z : w.x, w1 = c ? w.w : w.y;
                const int qpc = static_cast<int>(c ? (h1 & 255u) : (h0 >> 24));
                const int c0 = static_cast<int16_t>(w0 & 0xFFFFu), c1 = static_cast<int>(w0) >> 16, c2 = static_cast<int16_t>(w1 & 0xFFFFu), c3 = static_cast<int>(w1) >> 16;
                const int f = (c0 + ((blk & 2) ? -c2 : c2)) + ((blk & 1) ? -1 : 1) * (c1 + ((blk & 2) ? -c3 : c3)); // 8.5.11.1: c0 +- c1 +- c2 +- c3
                const int r = ((((f * sc->ls4[4 + c][qpc % 6][0]) << (qpc / 6)) >> 5) + 32) >> 6;
                cr[c][0] = cr[c][1] = cr[c][2] = cr[c][3] = r;
            }
        }
    } else if (__builtin_amdgcn_ballot_w64(has_c) != 0) {
        if (has_c && b < 8) {
            const int c = b >> 2, blk = b & 3;
            const int qpc = static_cast<int>(c ? (h1 & 255u) : (h0 >> 24));
            const uint16_t *ls = sc->ls4[4 + c][qpc % 6]; // inter Cb / Cr
            int d[16];
#pragma unroll
            for (int i = 0; i < 16; i++) d[i] = 0;
            const uint32_t ab = MI_COEF_CAC / 16 + c * 4 + blk;
            if ((cbp_c & 2) && ((cmask >> ab) & 1u)) {
                int cc[16];
                load_block(coefs, coef_off + __builtin_popcount(cmask & ((1u << ab) - 1u)), cc);
#pragma unroll
                for (int i = 1; i < 16; i++) d[i] = scale4(cc[i], ls[i], qpc);
            }
            const uint32_t db = MI_COEF_CDC / 16;
            if ((cmask >> db) & 1u) { // 8.5.11: 2x2 transform of the plane's DC coefficients
                const uint32_t *dcw = reinterpret_cast<const uint32_t *>(coefs) + 8 * static_cast<size_t>(coef_off + __builtin_popcount(cmask & ((1u << db) - 1u))) + 2 * c;
                const uint32_t w0 = dcw[0], w1 = dcw[1];
                const int c0 = static_cast<int16_t>(w0 & 0xFFFFu), c1 = static_cast<int>(w0) >> 16, c2 = static_cast<int16_t>(w1 & 0xFFFFu), c3 = static_cast<int>(w1) >> 16;
                const int f = blk == 0 ? c0 + c1 + c2 + c3 : (blk == 1 ? c0 - c1 + c2 - c3 : (blk == 2 ? c0 + c1 - c2 - c3 : c0 - c1 - c2 + c3));
                d[0] = ((f * ls[0]) << (qpc / 6)) >> 5;
            }
            int t[16];
#pragma unroll
            for (int r = 0; r < 4; r++) inv4(d[4 * r], d[4 * r + 1], d[4 * r + 2], d[4 * r + 3], t[4 * r], t[4 * r + 1], t[4 * r + 2], t[4 * r + 3]);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int o0, o1, o2, o3;
                inv4(t[k], t[4 + k], t[8 + k], t[12 + k], o0, o1, o2, o3);
                int16_t *dst = &ml->cres[c][((blk >> 1) * 4) * 8 + (blk & 1) * 4 + k];
                dst[0] = static_cast<int16_t>((o0 + 32) >> 6), dst[8] = static_cast<int16_t>((o1 + 32) >> 6), dst[16] = static_cast<int16_t>((o2 + 32) >> 6),
                dst[24] = static_cast<int16_t>((o3 + 32) >> 6);
            }
        }
        WAVE_SYNC();
        if (has_c) {
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const uint32_t w = *reinterpret_cast<const uint32_t *>(&ml->cres[c][(2 * by + j) * 8 + 2 * bx]);
                    cr[c][2 * j] = static_cast<int16_t>(w & 0xFFFFu), cr[c][2 * j + 1] = static_cast<int>(w) >> 16;
                }
        }
    }
    // ---- reconstruct + store ----
    if (inter) {
        g8 *dst = (g8 *)(pd->pool_base + static_cast<uint64_t>(pd->slot) * pd->slot_bytes) + (fld == 2 ? fpitch : 0u);
        const bool any_l = has_l || has_8;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t v = P[r];
            if (any_l) v = add_clip4(v, res[4 * r], res[4 * r + 1], res[4 * r + 2], res[4 * r + 3]);
            *reinterpret_cast<g32 *>(dst + (static_cast<uint32_t>(py + r) * pitch + static_cast<uint32_t>(px))) = v;
        }
        const uint32_t Wc = pitch >> 1;
        g8 *cdst = (g8 *)(pd->pool_base + static_cast<uint64_t>(pd->slot) * pd->slot_bytes) + plane + (fld == 2 ? fpitch >> 1 : 0u);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            uint32_t v = C[c];
            if (has_c) v = add_clip4(v, cr[c][0], cr[c][1], cr[c][2], cr[c][3]);
            g8 *cp = cdst + (c ? cr_delta : 0u) + (static_cast<uint32_t>(py >> 1) * Wc + static_cast<uint32_t>(px >> 1));
            *reinterpret_cast<g16 *>(cp) = static_cast<uint16_t>(v & 0xFFFFu);
            *reinterpret_cast<g16 *>(cp + Wc) = static_cast<uint16_t>(v >> 16);
        }
    }
}

// Register budget: the code needs 84 / 90 VGPRs (5 wavefronts per SIMD).  -DMI_K4_WAVES=6 forces 80: four registers spill, a launch of
// 256 pictures takes 3 % less (1.02 -> 0.99 ms: the short wavefronts are latency-bound) but moves 0.47 GB of scratch traffic on top of
// its 2.03 GB -- not worth it (tools/variant_k4.sh).
#ifdef MI_K4_WAVES
#define MI_K4_OCC __attribute__((amdgpu_waves_per_eu(MI_K4_WAVES, MI_K4_WAVES)))
#else
#define MI_K4_OCC
#endif
extern "C" __global__ void __launch_bounds__(64 * MI_K4_WG) MI_K4_OCC k_inter(const uint32_t *pic_list, const PicDesc *pics, const SliceDesc *slices, const DevTables *tab, const MbRec *mbrec,
                                                         const int16_t *coefs, int groups_per_pic_log2, int n_blocks) {
    __shared__ InterLds lds[4 * MI_K4_WG];
    inter4<false>(lds, pic_list, pics, slices, tab, mbrec, coefs, groups_per_pic_log2, n_blocks, nullptr, nullptr);
}
// K4 for the pictures that have B slices: two lists per block (MbRec::refslot1, MbMv1), default / explicit / implicit weighting
extern "C" __global__ void __launch_bounds__(64 * MI_K4_WG) MI_K4_OCC k_inter_b(const uint32_t *pic_list, const PicDesc *pics, const SliceDesc *slices, const DevTables *tab, const MbRec *mbrec,
                                                           const int16_t *coefs, int groups_per_pic_log2, int n_blocks, const BSliceExt *bexts, const MbMv1 *mbmv1) {
    __shared__ InterLds lds[4 * MI_K4_WG];
    inter4<true>(lds, pic_list, pics, slices, tab, mbrec, coefs, groups_per_pic_log2, n_blocks, bexts, mbmv1);
}

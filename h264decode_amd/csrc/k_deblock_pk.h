// h264decode_amd/csrc/k_deblock_pk.h -- the edge filters of 8.7.2.3 / 8.7.2.4 on TWO lines at once: a 32-bit register holds the same
// sample position of two lines (luma: two adjacent rows or columns of one 4-sample segment; chroma: the Cb and the Cr sample of one
// position), 16 bits each, and every operation is a packed 16-bit instruction (v_pk_add_u16 / v_pk_sub_i16 / v_pk_max_i16 /
// v_pk_min_i16 / v_pk_mad_i16 / v_pk_ashrrev_i16; conditions are sign masks, selects are v_bfi_b32, the rounding average is v_lerp_u8).
// The four edges of a line are a dependency chain (edge 8 reads what edge 4 wrote), so the two halves have to be two LINES, not two edges.
// Absent from the reference (only the slice-header fields are parsed: h264/slice.go:1021-1027); the arithmetic is 8.7.2.3 / 8.7.2.4 as written in the standard.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef short pk2 __attribute__((ext_vector_type(2)));
static_assert(sizeof(pk2) == 4, "two 16-bit halves");

__device__ __forceinline__ uint32_t pk_bits(pk2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ pk2 pk_from(uint32_t v) { return __builtin_bit_cast(pk2, v); }
__device__ __forceinline__ pk2 pk_splat(int v) { return pk2{static_cast<short>(v), static_cast<short>(v)}; }
__device__ __forceinline__ pk2 pk_max(pk2 a, pk2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ pk2 pk_min(pk2 a, pk2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ pk2 pk_absdiff(pk2 a, pk2 b) { return pk_max(a - b, b - a); }
// all ones in a half whose value is negative
__device__ __forceinline__ uint32_t pk_neg_mask(pk2 v) { return pk_bits(v >> pk_splat(15)); }
// m ? a : b, bit by bit (m is all ones or all zeros per half)
__device__ __forceinline__ pk2 pk_sel(uint32_t m, pk2 a, pk2 b) { return pk_from((pk_bits(a) & m) | (pk_bits(b) & ~m)); }
__device__ __forceinline__ pk2 pk_clip255(pk2 v) { return pk_min(pk_max(v, pk_splat(0)), pk_splat(255)); }
// (a + b + 1) >> 1 of samples 0..255 in the low byte of each half: one v_lerp_u8
__device__ __forceinline__ pk2 pk_avg(pk2 a, pk2 b) { return pk_from(__builtin_amdgcn_lerp(pk_bits(a), pk_bits(b), 0x00010001u)); }

// Luma edge of two lines.  alpha / beta / tc0: the edge's parameters in both halves; lane_on: all ones if bS > 0 for this lane's
// segment, else 0; lane_strong (MBEDGE only): all ones if bS == 4.  p3 / q3 are only read.  Returns false if no lane of the
// wavefront filtered anything (the caller may skip writing the line back).
template <bool MBEDGE>
__device__ __forceinline__ bool pk_luma_edge(pk2 p3, pk2 &p2, pk2 &p1, pk2 &p0, pk2 &q0, pk2 &q1, pk2 &q2, pk2 q3, pk2 alpha, pk2 beta, pk2 tc0, uint32_t lane_on,
                                             uint32_t lane_strong) {
    const pk2 dq = q0 - p0;
    const pk2 d0 = pk_max(dq, p0 - q0);
    const pk2 x = pk_max(pk_max(d0 - alpha, pk_absdiff(p1, p0) - beta), pk_absdiff(q1, q0) - beta);
    const uint32_t onm = pk_neg_mask(x) & lane_on;
    if (__builtin_amdgcn_ballot_w64(onm != 0) == 0) return false;
    const uint32_t apm = pk_neg_mask(pk_absdiff(p2, p0) - beta), aqm = pk_neg_mask(pk_absdiff(q2, q0) - beta);
    const uint32_t nm = MBEDGE ? onm & ~lane_strong : onm;
    {
        const pk2 tc = tc0 - pk_from(apm) - pk_from(aqm); // a true mask is -1
        pk2 v = dq * pk_splat(4) + (p1 - q1);
        v = (v + pk_splat(4)) >> pk_splat(3);
        const pk2 delta = pk_min(pk_max(v, pk_splat(0) - tc), tc);
        const pk2 avg = pk_avg(p0, q0), ntc0 = pk_splat(0) - tc0;
        pk2 tp = (p1 * pk_splat(-2) + (p2 + avg)) >> pk_splat(1);
        pk2 tq = (q1 * pk_splat(-2) + (q2 + avg)) >> pk_splat(1);
        tp = pk_min(pk_max(tp, ntc0), tc0) + p1;
        tq = pk_min(pk_max(tq, ntc0), tc0) + q1;
        const pk2 np0 = pk_clip255(p0 + delta), nq0 = pk_clip255(q0 - delta);
        if (MBEDGE && __builtin_amdgcn_ballot_w64((onm & lane_strong) != 0) != 0) {
            const uint32_t sm = onm & lane_strong;
            const uint32_t sms = sm & pk_neg_mask(d0 - ((alpha >> pk_splat(2)) + pk_splat(2)));
            const uint32_t spm = sms & apm, sqm = sms & aqm;
            const pk2 t = p0 + q0, s3 = t + p1, t3 = t + q1;
            const pk2 P0s = (s3 * pk_splat(2) + p2 + q1 + pk_splat(4)) >> pk_splat(3);
            const pk2 P1s = (p2 + s3 + pk_splat(2)) >> pk_splat(2);
            const pk2 P2s = (p3 * pk_splat(2) + p2 * pk_splat(3) + s3 + pk_splat(4)) >> pk_splat(3);
            const pk2 P0w = (p1 * pk_splat(2) + p0 + q1 + pk_splat(2)) >> pk_splat(2);
            const pk2 Q0s = (t3 * pk_splat(2) + q2 + p1 + pk_splat(4)) >> pk_splat(3);
            const pk2 Q1s = (q2 + t3 + pk_splat(2)) >> pk_splat(2);
            const pk2 Q2s = (q3 * pk_splat(2) + q2 * pk_splat(3) + t3 + pk_splat(4)) >> pk_splat(3);
            const pk2 Q0w = (q1 * pk_splat(2) + q0 + p1 + pk_splat(2)) >> pk_splat(2);
            const pk2 op0 = p0, oq0 = q0;
            p0 = pk_sel(spm, P0s, pk_sel(sm, P0w, pk_sel(nm, np0, op0)));
            q0 = pk_sel(sqm, Q0s, pk_sel(sm, Q0w, pk_sel(nm, nq0, oq0)));
            p1 = pk_sel(spm, P1s, pk_sel(nm & apm, tp, p1));
            q1 = pk_sel(sqm, Q1s, pk_sel(nm & aqm, tq, q1));
            p2 = pk_sel(spm, P2s, p2);
            q2 = pk_sel(sqm, Q2s, q2);
            return true;
        }
        p0 = pk_sel(nm, np0, p0), q0 = pk_sel(nm, nq0, q0);
        p1 = pk_sel(nm & apm, tp, p1), q1 = pk_sel(nm & aqm, tq, q1);
    }
    return true;
}

// Chroma edge of two lines (the Cb and the Cr line of one position): tc = tC0 + 1 for bS < 4 already added by the caller.
template <bool MBEDGE>
__device__ __forceinline__ bool pk_chroma_edge(pk2 p1, pk2 &p0, pk2 &q0, pk2 q1, pk2 alpha, pk2 beta, pk2 tc, uint32_t lane_on, uint32_t lane_strong) {
    const pk2 dq = q0 - p0;
    const pk2 d0 = pk_max(dq, p0 - q0);
    const pk2 x = pk_max(pk_max(d0 - alpha, pk_absdiff(p1, p0) - beta), pk_absdiff(q1, q0) - beta);
    const uint32_t onm = pk_neg_mask(x) & lane_on;
    if (__builtin_amdgcn_ballot_w64(onm != 0) == 0) return false;
    pk2 v = dq * pk_splat(4) + (p1 - q1);
    v = (v + pk_splat(4)) >> pk_splat(3);
    const pk2 delta = pk_min(pk_max(v, pk_splat(0) - tc), tc);
    pk2 np0 = pk_clip255(p0 + delta), nq0 = pk_clip255(q0 - delta);
    if (MBEDGE && __builtin_amdgcn_ballot_w64((onm & lane_strong) != 0) != 0) {
        const uint32_t sm = onm & lane_strong;
        np0 = pk_sel(sm, (p1 * pk_splat(2) + p0 + q1 + pk_splat(2)) >> pk_splat(2), np0);
        nq0 = pk_sel(sm, (q1 * pk_splat(2) + q0 + p1 + pk_splat(2)) >> pk_splat(2), nq0);
    }
    p0 = pk_sel(onm, np0, p0), q0 = pk_sel(onm, nq0, q0);
    return true;
}

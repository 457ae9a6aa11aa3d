// The packed two-lines-per-lane edge filters of K5 (h264decode_amd/csrc/k_deblock_pk.h) against a plain scalar statement of 8.7.2.3 / 8.7.2.4
// on random lines and parameters (bring-up aid for the K5 rewrite of round 5; bit-exactness + the ISA instruction count).
//   hipcc --offload-arch=gfx950 -O3 -I../../h264decode_amd/csrc pk_filter.hip -o pk_filter && ./pk_filter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include "k_deblock_pk.h"

struct Line {
    uint8_t s[2][8];   // two lines, p3 p2 p1 p0 q0 q1 q2 q3
    uint8_t alpha, beta, tc0, bs;
};

static inline int clip3h(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static void ref_luma(uint8_t *s, int bs, int alpha, int beta, int tc0) {
    const int p3 = s[0], p2 = s[1], p1 = s[2], p0 = s[3], q0 = s[4], q1 = s[5], q2 = s[6], q3 = s[7];
    if (!bs || !(abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta)) return;
    const bool ap = abs(p2 - p0) < beta, aq = abs(q2 - q0) < beta;
    if (bs < 4) {
        const int tc = tc0 + ap + aq;
        const int delta = clip3h(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        s[3] = (uint8_t)clip3h(0, 255, p0 + delta), s[4] = (uint8_t)clip3h(0, 255, q0 - delta);
        if (ap) s[2] = (uint8_t)(p1 + clip3h(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1));
        if (aq) s[5] = (uint8_t)(q1 + clip3h(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1));
    } else {
        const bool small = abs(p0 - q0) < ((alpha >> 2) + 2);
        if (ap && small) {
            s[3] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3), s[2] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
            s[1] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
        } else
            s[3] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        if (aq && small) {
            s[4] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3), s[5] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
            s[6] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
        } else
            s[4] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    }
}
static void ref_chroma(uint8_t *s, int bs, int alpha, int beta, int tc0) {
    const int p1 = s[2], p0 = s[3], q0 = s[4], q1 = s[5];
    if (!bs || !(abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta)) return;
    if (bs < 4) {
        const int tc = tc0 + 1;
        const int delta = clip3h(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        s[3] = (uint8_t)clip3h(0, 255, p0 + delta), s[4] = (uint8_t)clip3h(0, 255, q0 - delta);
    } else
        s[3] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2), s[4] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
}

template <bool MBEDGE, bool CHROMA>
__global__ void k_pk(const Line *in, Line *out, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Line l = in[t];
    pk2 r[8];
    for (int i = 0; i < 8; i++) r[i] = pk2{static_cast<short>(l.s[0][i]), static_cast<short>(l.s[1][i])};
    const uint32_t on = l.bs ? ~0u : 0u, strong = l.bs == 4 ? ~0u : 0u;
    if (CHROMA)
        pk_chroma_edge<MBEDGE>(r[2], r[3], r[4], r[5], pk_splat(l.alpha), pk_splat(l.beta), pk_splat(l.tc0 + 1), on, strong);
    else
        pk_luma_edge<MBEDGE>(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], pk_splat(l.alpha), pk_splat(l.beta), pk_splat(l.tc0), on, strong);
    for (int i = 0; i < 8; i++) l.s[0][i] = static_cast<uint8_t>(r[i].x), l.s[1][i] = static_cast<uint8_t>(r[i].y);
    out[t] = l;
}

int main() {
    const int N = 1 << 20;
    Line *h = (Line *)malloc(N * sizeof(Line)), *o = (Line *)malloc(N * sizeof(Line)), *d_in, *d_out;
    static const uint8_t alpha_t[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
    static const uint8_t beta_t[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
    int total_bad = 0;
    (void)hipMalloc(&d_in, N * sizeof(Line)), (void)hipMalloc(&d_out, N * sizeof(Line));
    for (int variant = 0; variant < 4; variant++) { // luma inner / luma macroblock edge / chroma inner / chroma macroblock edge
        const bool mbedge = variant & 1, chroma = variant >= 2;
        srand(7 + variant);
        for (int i = 0; i < N; i++) {
            Line &l = h[i];
            const int mode = rand() % 4; // flat-ish lines (so that the filters switch on), steps, noise, extremes
            const int base = rand() % 256, amp = mode == 0 ? 3 : (mode == 1 ? 12 : (mode == 2 ? 60 : 255));
            for (int k = 0; k < 2; k++)
                for (int j = 0; j < 8; j++) {
                    int v = base + (amp ? rand() % (2 * amp + 1) - amp : 0) + ((mode == 1 && j >= 4) ? rand() % 24 - 12 : 0);
                    if (mode == 3) v = (rand() & 1) ? (rand() & 1 ? 255 : 0) : v;
                    l.s[k][j] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                }
            const int q = rand() % 52;
            l.alpha = alpha_t[q], l.beta = beta_t[rand() % 3 ? q : rand() % 52], l.tc0 = (uint8_t)(rand() % 26);
            l.bs = (uint8_t)(rand() % (mbedge ? 5 : 4));
        }
        (void)hipMemcpy(d_in, h, N * sizeof(Line), hipMemcpyHostToDevice);
        if (variant == 0) k_pk<false, false><<<N / 256, 256>>>(d_in, d_out, N);
        if (variant == 1) k_pk<true, false><<<N / 256, 256>>>(d_in, d_out, N);
        if (variant == 2) k_pk<false, true><<<N / 256, 256>>>(d_in, d_out, N);
        if (variant == 3) k_pk<true, true><<<N / 256, 256>>>(d_in, d_out, N);
        (void)hipMemcpy(o, d_out, N * sizeof(Line), hipMemcpyDeviceToHost);
        int bad = 0, changed = 0;
        for (int i = 0; i < N; i++) {
            Line w = h[i];
            for (int k = 0; k < 2; k++) chroma ? ref_chroma(w.s[k], w.bs, w.alpha, w.beta, w.tc0) : ref_luma(w.s[k], w.bs, w.alpha, w.beta, w.tc0);
            if (memcmp(w.s, h[i].s, 16)) changed++;
            if (memcmp(w.s, o[i].s, 16) && bad++ < 4) {
                printf("variant %d line %d bs %d alpha %d beta %d tc0 %d\n", variant, i, w.bs, w.alpha, w.beta, w.tc0);
                for (int k = 0; k < 2; k++) {
                    printf("  in  "); for (int j = 0; j < 8; j++) printf("%4d", h[i].s[k][j]);
                    printf("\n  ref "); for (int j = 0; j < 8; j++) printf("%4d", w.s[k][j]);
                    printf("\n  got "); for (int j = 0; j < 8; j++) printf("%4d", o[i].s[k][j]);
                    printf("\n");
                }
            }
        }
        printf("variant %d (%s %s): %d mismatches of %d line pairs, %d pairs changed by the filter\n", variant, chroma ? "chroma" : "luma", mbedge ? "macroblock edge" : "inner edge", bad, N, changed);
        total_bad += bad;
    }
    printf(total_bad ? "FAIL\n" : "PASS\n");
    return total_bad != 0;
}

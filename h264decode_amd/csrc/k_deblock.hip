// h264decode_amd/csrc/k_deblock.hip -- K5: in-loop deblocking filter (ITU-T H.264 8.7), gfx950.
//
// 8.7 is specified per macroblock in raster order (vertical edges left to right, then horizontal
// edges top to bottom), and the left-edge filter of MB(x+1,y) rewrites columns of MB(x,y) AFTER
// MB(x,y)'s horizontal edges were filtered, so a whole-picture "all vertical, then all horizontal"
// pass is not bit-exact.  The exact dependency is MB(x,y) after MB(x-1,y) and MB(x+1,y-1): a 2-D
// wavefront.  One workgroup owns a picture (no cross-CU hand-off), wavefront w owns macroblock rows
// w, w+16, ... and waits on the LDS progress counter of the row above.  Per macroblock the 20x20 luma
// and two 10x12 chroma neighbourhoods are staged in an LDS tile (dword loads), lanes 0-15 filter
// luma rows / columns and lanes 16-31 chroma, and the modified samples go back with dword stores.
//
// Absent from the reference (only the slice-header fields are parsed: h264/slice.go:1021-1027).
#include <hip/hip_runtime.h>
#include "mi_kernels.h"

#define WAVE_SYNC()                                             \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)

struct DbWave {
    uint8_t y[20][20];     // rows/cols -4..15 of the macroblock
    uint8_t c[2][12][12];  // rows -4..7 (only -2.. used), cols -4..7
    uint8_t bs[2][4][4];   // [dir][edge][segment]
    uint8_t any[2];
};
struct DbShared {
    DbWave w[MI_DEBLOCK_WAVES];
    int prog[320];
};

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }

// filter one line of samples; p points at q0, `step` = distance between samples across the edge
__device__ __forceinline__ void filter_line(uint8_t *q0p, int step, int bs, int alpha, int beta, int tc0, bool chroma) {
    int p0 = q0p[-step], p1 = q0p[-2 * step], q0 = q0p[0], q1 = q0p[step];
    if (!(iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta)) return;
    if (bs < 4) {
        int tc;
        if (chroma)
            tc = tc0 + 1;
        else {
            int p2 = q0p[-3 * step], q2 = q0p[2 * step];
            int ap = iabs(p2 - p0), aq = iabs(q2 - q0);
            tc = tc0 + (ap < beta) + (aq < beta);
            if (ap < beta) q0p[-2 * step] = static_cast<uint8_t>(p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1));
            if (aq < beta) q0p[step] = static_cast<uint8_t>(q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1));
        }
        int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        q0p[-step] = static_cast<uint8_t>(clip3(0, 255, p0 + delta));
        q0p[0] = static_cast<uint8_t>(clip3(0, 255, q0 - delta));
    } else if (chroma) {
        q0p[-step] = static_cast<uint8_t>((2 * p1 + p0 + q1 + 2) >> 2);
        q0p[0] = static_cast<uint8_t>((2 * q1 + q0 + p1 + 2) >> 2);
    } else {
        int p2 = q0p[-3 * step], q2 = q0p[2 * step];
        int ap = iabs(p2 - p0), aq = iabs(q2 - q0);
        bool small = iabs(p0 - q0) < ((alpha >> 2) + 2);
        if (ap < beta && small) {
            int p3 = q0p[-4 * step];
            q0p[-step] = static_cast<uint8_t>((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            q0p[-2 * step] = static_cast<uint8_t>((p2 + p1 + p0 + q0 + 2) >> 2);
            q0p[-3 * step] = static_cast<uint8_t>((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
        } else
            q0p[-step] = static_cast<uint8_t>((2 * p1 + p0 + q1 + 2) >> 2);
        if (aq < beta && small) {
            int q3 = q0p[3 * step];
            q0p[0] = static_cast<uint8_t>((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
            q0p[step] = static_cast<uint8_t>((p0 + q0 + q1 + q2 + 2) >> 2);
            q0p[2 * step] = static_cast<uint8_t>((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
        } else
            q0p[0] = static_cast<uint8_t>((2 * q1 + q0 + p1 + 2) >> 2);
    }
}

// 8.7.2.1 for P/I frame macroblocks
__device__ __forceinline__ int edge_bs(const MbRec *mp, int pb, const MbRec *mq, int qb, bool mb_edge) {
    if (MB_IS_INTRA(mp->type) || MB_IS_INTRA(mq->type)) return mb_edge ? 4 : 3;
    if (((mp->nzmask >> pb) & 1) || ((mq->nzmask >> qb) & 1)) return 2;
    int rp = mp->refslot[((pb >> 3) << 1) | ((pb & 3) >> 1)], rq = mq->refslot[((qb >> 3) << 1) | ((qb & 3) >> 1)];
    if (rp != rq) return 1;
    if (iabs(mp->mv[pb][0] - mq->mv[qb][0]) >= 4 || iabs(mp->mv[pb][1] - mq->mv[qb][1]) >= 4) return 1;
    return 0;
}

__device__ void deblock_mb(int lane, DbWave *ws, const MbRec *mq, const MbRec *mleft, const MbRec *mtop, const DevTables *tab, uint8_t *py, uint8_t *pcb,
                           uint8_t *pcr, int W, int mbx, int mby) {
    // ---- boundary strengths: lanes 0..31 = (dir, edge, segment) ----
    if (lane < 2) ws->any[lane] = 0;
    WAVE_SYNC();
    if (lane < 32) {
        const int dir = lane >> 4, e = (lane >> 2) & 3, k = lane & 3;
        const MbRec *mn = dir == 0 ? mleft : mtop;
        int bs = 0;
        if (!(e == 0 && !mn) && !((e & 1) && mq->t8x8)) {
            const MbRec *mp = e == 0 ? mn : mq;
            int qb = dir == 0 ? k * 4 + e : e * 4 + k;
            int pb = dir == 0 ? k * 4 + (e == 0 ? 3 : e - 1) : (e == 0 ? 3 : e - 1) * 4 + k;
            bs = edge_bs(mp, pb, mq, qb, e == 0);
        }
        ws->bs[dir][e][k] = static_cast<uint8_t>(bs);
        if (bs) ws->any[dir] = 1;
    }
    WAVE_SYNC();
    if (!(ws->any[0] | ws->any[1])) return;
    const int Wc = W / 2;
    uint8_t *Y = py + static_cast<size_t>(mby * 16) * W + mbx * 16;
    uint8_t *C[2] = {pcb + static_cast<size_t>(mby * 8) * Wc + mbx * 8, pcr + static_cast<size_t>(mby * 8) * Wc + mbx * 8};
    const bool has_left = mbx > 0, has_top = mby > 0;
    // ---- stage the neighbourhood in LDS: luma 20 rows x 5 dwords, chroma 2 x 12 rows x 3 dwords ----
    for (int i = lane; i < 100; i += 64) {
        int r = i / 5, d = i - r * 5;
        uint32_t v = 0;
        if ((r >= 4 || has_top) && (d >= 1 || has_left)) v = *reinterpret_cast<const uint32_t *>(Y + static_cast<ptrdiff_t>(r - 4) * W + (d - 1) * 4);
        *reinterpret_cast<uint32_t *>(&ws->y[r][d * 4]) = v;
    }
    for (int i = lane; i < 72; i += 64) {
        int c = i / 36, rem = i - c * 36, r = rem / 3, d = rem - r * 3;
        uint32_t v = 0;
        if ((r >= 4 || has_top) && (d >= 1 || has_left)) v = *reinterpret_cast<const uint32_t *>(C[c] + static_cast<ptrdiff_t>(r - 4) * Wc + (d - 1) * 4);
        *reinterpret_cast<uint32_t *>(&ws->c[c][r][d * 4]) = v;
    }
    WAVE_SYNC();
    // ---- the two filtering passes ----
    for (int dir = 0; dir < 2; dir++) {
        if (ws->any[dir]) {
            const MbRec *mn = dir == 0 ? mleft : mtop;
            if (lane < 16) { // luma: one row (dir 0) or column (dir 1) per lane, edges in order
                for (int e = 0; e < 4; e++) {
                    int bs = ws->bs[dir][e][lane >> 2];
                    if (!bs) continue;
                    const MbRec *mp = e == 0 ? mn : mq;
                    int qpav = (mp->qp + mq->qp + 1) >> 1;
                    int ia = clip3(0, 51, qpav + mq->alpha_off), ib = clip3(0, 51, qpav + mq->beta_off);
                    uint8_t *q0 = dir == 0 ? &ws->y[4 + lane][4 + e * 4] : &ws->y[4 + e * 4][4 + lane];
                    filter_line(q0, dir == 0 ? 1 : 20, bs, tab->alpha[ia], tab->beta[ib], bs < 4 ? tab->tc0[ia][bs] : 0, false);
                }
            } else if (lane < 32) { // chroma: plane = bit 3, row/column = low 3 bits; luma edges 0 and 2
                const int c = (lane >> 3) & 1, i = lane & 7;
                for (int e = 0; e < 4; e += 2) {
                    int bs = ws->bs[dir][e][i >> 1];
                    if (!bs) continue;
                    const MbRec *mp = e == 0 ? mn : mq;
                    int qpav = (mp->qpc[c] + mq->qpc[c] + 1) >> 1;
                    int ia = clip3(0, 51, qpav + mq->alpha_off), ib = clip3(0, 51, qpav + mq->beta_off);
                    uint8_t *q0 = dir == 0 ? &ws->c[c][4 + i][4 + e * 2] : &ws->c[c][4 + e * 2][4 + i];
                    filter_line(q0, dir == 0 ? 1 : 12, bs, tab->alpha[ia], tab->beta[ib], bs < 4 ? tab->tc0[ia][bs] : 0, true);
                }
            }
        }
        WAVE_SYNC();
    }
    // ---- write back rows -3..15 (all 5 dwords) ----
    for (int i = lane; i < 100; i += 64) {
        int r = i / 5, d = i - r * 5;
        if (r >= 1 && (r >= 4 || has_top) && (d >= 1 || has_left) && !(r < 4 && d == 0)) // the top-left corner block is never modified here
            *reinterpret_cast<uint32_t *>(Y + static_cast<ptrdiff_t>(r - 4) * W + (d - 1) * 4) = *reinterpret_cast<const uint32_t *>(&ws->y[r][d * 4]);
    }
    for (int i = lane; i < 72; i += 64) {
        int c = i / 36, rem = i - c * 36, r = rem / 3, d = rem - r * 3;
        if (r >= 3 && (r >= 4 || has_top) && (d >= 1 || has_left) && !(r < 4 && d == 0))
            *reinterpret_cast<uint32_t *>(C[c] + static_cast<ptrdiff_t>(r - 4) * Wc + (d - 1) * 4) = *reinterpret_cast<const uint32_t *>(&ws->c[c][r][d * 4]);
    }
}

extern "C" __global__ void __launch_bounds__(MI_DEBLOCK_WAVES * 64) k_deblock(const uint32_t *pic_list, const PicDesc *pics, const FramePool *pools,
                                                                              const DevTables *tab, const MbRec *mbrec) {
    __shared__ DbShared sh;
    const int tid = static_cast<int>(threadIdx.x), lane = tid & 63, wave = tid >> 6;
    const PicDesc *pd = &pics[pic_list[blockIdx.x]];
    const int wmb = static_cast<int>(pd->wmb), hmb = static_cast<int>(pd->hmb);
    const FramePool *pool = &pools[pd->stream];
    const int W = static_cast<int>(pool->w), H = static_cast<int>(pool->h);
    uint8_t *py = reinterpret_cast<uint8_t *>(pool->base) + static_cast<size_t>(pd->slot) * pool->slot_bytes;
    uint8_t *pcb = py + static_cast<size_t>(W) * H, *pcr = pcb + static_cast<size_t>(W) * H / 4;
    for (int i = tid; i < 320; i += MI_DEBLOCK_WAVES * 64) sh.prog[i] = 0;
    __syncthreads();
    DbWave *ws = &sh.w[wave];
    for (int mby = wave; mby < hmb; mby += MI_DEBLOCK_WAVES) {
        for (int mbx = 0; mbx < wmb; mbx++) {
            const MbRec *mq = mbrec + pd->mb_base + static_cast<uint64_t>(mby) * wmb + mbx;
            if (mq->dbf_idc != 1) {
                const MbRec *ml = mbx > 0 ? mq - 1 : nullptr, *mt = mby > 0 ? mq - wmb : nullptr;
                if (mq->dbf_idc == 2) { // no filtering across slice boundaries
                    if (ml && ml->slice_in_pic != mq->slice_in_pic) ml = nullptr;
                    if (mt && mt->slice_in_pic != mq->slice_in_pic) mt = nullptr;
                }
                if (mby > 0) {
                    const int need = min(mbx + 2, wmb);
                    while (__hip_atomic_load(&sh.prog[mby - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
                }
                deblock_mb(lane, ws, mq, ml, mt, tab, py, pcb, pcr, W, mbx, mby);
            }
            if (lane == 0) __hip_atomic_store(&sh.prog[mby], mbx + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

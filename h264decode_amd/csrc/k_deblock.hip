// h264decode_amd/csrc/k_deblock.hip -- K5: in-loop deblocking filter (ITU-T H.264 8.7), gfx950.
//
// 8.7 is specified per macroblock in raster order (vertical edges left to right, then horizontal
// edges top to bottom), and the left-edge filter of MB(x+1,y) rewrites columns of MB(x,y) AFTER
// MB(x,y)'s horizontal edges were filtered, so a whole-picture "all vertical, then all horizontal"
// pass is not bit-exact.  The exact dependency is MB(x,y) after MB(x-1,y) and MB(x+1,y-1): a 2-D
// wavefront.  One workgroup owns a picture (no cross-CU hand-off), wavefront w owns macroblock rows
// w, w+16, ... and waits on the LDS progress counter of the row above.
//
// Per macroblock (one wavefront):
//   * the MbRec of the current / left / upper macroblock and the alpha/beta/tC0 tables live in LDS,
//     so no filtering decision ever waits on a dependent global load;
//   * the macroblock's own 16x16 + 2x 8x8 samples and its MbRecs are PREFETCHED into registers one
//     macroblock ahead (they are final inputs from K3/K4); only the 4 rows above (written by the
//     wavefront of the previous row) are loaded after the progress wait, and the 4 columns to the
//     left are carried over inside LDS from the previous tile;
//   * lanes 0-15 filter luma rows / columns and lanes 16-31 chroma in a 20x20 / 12x12 LDS tile;
//     the result goes back with dword stores (rows -3..15, columns -4..15).
//
// Absent from the reference (only the slice-header fields are parsed: h264/slice.go:1021-1027).
#include <hip/hip_runtime.h>
#include "mi_kernels.h"

#define WAVE_SYNC()                                            \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

struct DbTile {
    uint8_t y[20][20];    // rows/cols -4..15 of the macroblock
    uint8_t c[2][12][12]; // rows -4..7 (only -2.. used), cols -4..7
};
struct DbWave {
    DbTile tile[2];     // double buffer: the left 4 columns of tile[k] come from tile[k^1]
    MbRec rec[3];       // ring: cur / left share slots (left = previous cur), top
    uint8_t bs[2][4][4]; // [dir][edge][segment]
    uint8_t any[2], pad[2];
};
struct DbShared {
    DbWave w[MI_DEBLOCK_WAVES];
    uint8_t alpha[52], beta[52], tc0[52][4];
    int prog[320];
};

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }

// filter one edge of a line of samples held in registers (8.7.2.3 / 8.7.2.4); q0 = px[Q]
template <int Q, bool CHROMA, int N>
__device__ __forceinline__ void filter_edge(int (&px)[N], int bs, int alpha, int beta, int tc0) {
    const int p0 = px[Q - 1], p1 = px[Q - 2], q0 = px[Q], q1 = px[Q + 1];
    if (!bs || !(iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta)) return;
    if (bs < 4) {
        int tc;
        if (CHROMA)
            tc = tc0 + 1;
        else {
            const int p2 = px[Q - 3], q2 = px[Q + 2];
            const int ap = iabs(p2 - p0), aq = iabs(q2 - q0);
            tc = tc0 + (ap < beta) + (aq < beta);
            if (ap < beta) px[Q - 2] = p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1);
            if (aq < beta) px[Q + 1] = q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1);
        }
        const int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        px[Q - 1] = clip3(0, 255, p0 + delta);
        px[Q] = clip3(0, 255, q0 - delta);
    } else if (CHROMA) {
        px[Q - 1] = (2 * p1 + p0 + q1 + 2) >> 2;
        px[Q] = (2 * q1 + q0 + p1 + 2) >> 2;
    } else {
        const int p2 = px[Q - 3], q2 = px[Q + 2];
        const int ap = iabs(p2 - p0), aq = iabs(q2 - q0);
        const bool small = iabs(p0 - q0) < ((alpha >> 2) + 2);
        if (ap < beta && small) {
            const int p3 = px[Q - 4];
            px[Q - 1] = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3;
            px[Q - 2] = (p2 + p1 + p0 + q0 + 2) >> 2;
            px[Q - 3] = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
        } else
            px[Q - 1] = (2 * p1 + p0 + q1 + 2) >> 2;
        if (aq < beta && small) {
            const int q3 = px[Q + 3];
            px[Q] = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3;
            px[Q + 1] = (p0 + q0 + q1 + q2 + 2) >> 2;
            px[Q + 2] = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
        } else
            px[Q] = (2 * q1 + q0 + p1 + 2) >> 2;
    }
}

// 8.7.2.1 for frame macroblocks of I/P pictures
__device__ __forceinline__ int edge_bs(const MbRec *mp, int pb, const MbRec *mq, int qb, bool mb_edge) {
    if (MB_IS_INTRA(mp->type) || MB_IS_INTRA(mq->type)) return mb_edge ? 4 : 3;
    if (((mp->nzmask >> pb) & 1) || ((mq->nzmask >> qb) & 1)) return 2;
    int rp = mp->refslot[((pb >> 3) << 1) | ((pb & 3) >> 1)], rq = mq->refslot[((qb >> 3) << 1) | ((qb & 3) >> 1)];
    if (rp != rq) return 1;
    if (iabs(mp->mv[pb][0] - mq->mv[qb][0]) >= 4 || iabs(mp->mv[pb][1] - mq->mv[qb][1]) >= 4) return 1;
    return 0;
}

extern "C" __global__ void __launch_bounds__(MI_DEBLOCK_WAVES * 64) k_deblock(const uint32_t *pic_list, const PicDesc *pics, const FramePool *pools,
                                                                              const DevTables *tab, const MbRec *mbrec) {
    __shared__ DbShared sh;
    const int tid = static_cast<int>(threadIdx.x), lane = tid & 63, wave = tid >> 6;
    const PicDesc *pd = &pics[pic_list[blockIdx.x]];
    const int wmb = static_cast<int>(pd->wmb), hmb = static_cast<int>(pd->hmb);
    const FramePool *pool = &pools[pd->stream];
    const int W = static_cast<int>(pool->w), H = static_cast<int>(pool->h), Wc = W / 2;
    uint8_t *py = reinterpret_cast<uint8_t *>(pool->base) + static_cast<size_t>(pd->slot) * pool->slot_bytes;
    uint8_t *pcb = py + static_cast<size_t>(W) * H, *pcr = pcb + static_cast<size_t>(W) * H / 4;
    for (int i = tid; i < 320; i += MI_DEBLOCK_WAVES * 64) sh.prog[i] = 0;
    for (int i = tid; i < 52; i += MI_DEBLOCK_WAVES * 64) {
        sh.alpha[i] = tab->alpha[i], sh.beta[i] = tab->beta[i];
        sh.tc0[i][0] = 0, sh.tc0[i][1] = tab->tc0[i][1], sh.tc0[i][2] = tab->tc0[i][2], sh.tc0[i][3] = tab->tc0[i][3];
    }
    __syncthreads();
    DbWave *ws = &sh.w[wave];
    const MbRec *recs = mbrec + pd->mb_base;
    for (int mby = wave; mby < hmb; mby += MI_DEBLOCK_WAVES) {
        const bool has_top = mby > 0;
        // ---- prefetch for mbx = 0: MbRecs (cur: lanes 0-31, top: lanes 32-63) and own samples ----
        const MbRec *row = recs + static_cast<size_t>(mby) * wmb;
        uint32_t pre_rec = 0, pre_y = 0, pre_c = 0;
        auto prefetch = [&](int mbx) {
            const MbRec *src = lane < 32 ? row + mbx : (has_top ? row + mbx - wmb : row + mbx);
            pre_rec = reinterpret_cast<const uint32_t *>(src)[lane & 31];
            // own 16x16 luma: 64 dwords (lane -> row lane/4, dword lane%4); own chroma: 2 x 8 rows x 2 dwords on lanes 0-31
            pre_y = *reinterpret_cast<const uint32_t *>(py + static_cast<size_t>(mby * 16 + (lane >> 2)) * W + mbx * 16 + (lane & 3) * 4);
            if (lane < 32) {
                const uint8_t *cp = (lane < 16 ? pcb : pcr) + static_cast<size_t>(mby * 8 + ((lane >> 1) & 7)) * Wc + mbx * 8 + (lane & 1) * 4;
                pre_c = *reinterpret_cast<const uint32_t *>(cp);
            }
        };
        prefetch(0);
        int cur_slot = 0; // rec ring: cur = rec[cur_slot], left = rec[cur_slot ^ 1], top = rec[2]
        for (int mbx = 0; mbx < wmb; mbx++) {
            cur_slot ^= 1;
            DbTile *tl = &ws->tile[mbx & 1], *prev = &ws->tile[(mbx & 1) ^ 1];
            MbRec *mq = &ws->rec[cur_slot], *mleft_rec = &ws->rec[cur_slot ^ 1], *mtop_rec = &ws->rec[2];
            // ---- commit the prefetched data to LDS ----
            reinterpret_cast<uint32_t *>(lane < 32 ? mq : mtop_rec)[lane & 31] = pre_rec;
            *reinterpret_cast<uint32_t *>(&tl->y[4 + (lane >> 2)][4 + (lane & 3) * 4]) = pre_y;
            if (lane < 32) *reinterpret_cast<uint32_t *>(&tl->c[lane >> 4][4 + ((lane >> 1) & 7)][4 + (lane & 1) * 4]) = pre_c;
            // left 4 columns: carried over from the previous tile (rows 0..15 luma, 0..7 chroma)
            if (mbx > 0) {
                if (lane < 16)
                    *reinterpret_cast<uint32_t *>(&tl->y[4 + lane][0]) = *reinterpret_cast<const uint32_t *>(&prev->y[4 + lane][16]);
                else if (lane < 32)
                    *reinterpret_cast<uint32_t *>(&tl->c[(lane >> 3) & 1][4 + (lane & 7)][0]) = *reinterpret_cast<const uint32_t *>(&prev->c[(lane >> 3) & 1][4 + (lane & 7)][8]);
            }
            if (lane < 2) ws->any[lane] = 0;
            WAVE_SYNC();
            if (mbx + 1 < wmb) prefetch(mbx + 1);
            const int dbf = mq->dbf_idc;
            if (dbf != 1) {
                const MbRec *ml = mbx > 0 ? mleft_rec : nullptr, *mt = has_top ? mtop_rec : nullptr;
                if (dbf == 2) { // no filtering across slice boundaries
                    if (ml && ml->slice_in_pic != mq->slice_in_pic) ml = nullptr;
                    if (mt && mt->slice_in_pic != mq->slice_in_pic) mt = nullptr;
                }
                // ---- boundary strengths: lanes 0..31 = (dir, edge, segment) ----
                if (lane < 32) {
                    const int dir = lane >> 4, e = (lane >> 2) & 3, k = lane & 3;
                    const MbRec *mn = dir == 0 ? ml : mt;
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
                if (ws->any[0] | ws->any[1]) {
                    uint8_t *Y = py + static_cast<size_t>(mby * 16) * W + mbx * 16;
                    uint8_t *C0 = pcb + static_cast<size_t>(mby * 8) * Wc + mbx * 8, *C1 = pcr + static_cast<size_t>(mby * 8) * Wc + mbx * 8;
                    // ---- the 4 rows above come from the wavefront of the previous row ----
                    if (has_top) {
                        const int need = min(mbx + 2, wmb);
                        while (__hip_atomic_load(&sh.prog[mby - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
                        if (lane < 16) // luma rows -4..-1, columns 0..15
                            *reinterpret_cast<uint32_t *>(&tl->y[lane >> 2][4 + (lane & 3) * 4]) =
                                *reinterpret_cast<const uint32_t *>(Y + static_cast<ptrdiff_t>((lane >> 2) - 4) * W + (lane & 3) * 4);
                        else if (lane < 24) { // chroma rows -2..-1
                            const int c = (lane >> 2) & 1, r = (lane >> 1) & 1, d = lane & 1;
                            *reinterpret_cast<uint32_t *>(&tl->c[c][2 + r][4 + d * 4]) =
                                *reinterpret_cast<const uint32_t *>((c ? C1 : C0) + static_cast<ptrdiff_t>(r - 2) * Wc + d * 4);
                        }
                        WAVE_SYNC();
                    }
                    // ---- the two filtering passes: a whole line of samples in registers per lane ----
                    for (int dir = 0; dir < 2; dir++) {
                        if (ws->any[dir]) {
                            const MbRec *mn = dir == 0 ? ml : mt;
                            if (lane < 16) { // luma: one row (dir 0) or column (dir 1) per lane
                                int px[20];
                                if (dir == 0) {
#pragma unroll
                                    for (int d = 0; d < 5; d++) {
                                        uint32_t w = *reinterpret_cast<const uint32_t *>(&tl->y[4 + lane][d * 4]);
                                        px[4 * d] = w & 255, px[4 * d + 1] = (w >> 8) & 255, px[4 * d + 2] = (w >> 16) & 255, px[4 * d + 3] = w >> 24;
                                    }
                                } else {
#pragma unroll
                                    for (int r = 0; r < 20; r++) px[r] = tl->y[r][4 + lane];
                                }
                                const uint32_t bsw = *reinterpret_cast<const uint32_t *>(ws->bs[dir][0]) >> (8 * (lane >> 2));
                                const int bs0 = bsw & 255;
                                const int bs1 = ws->bs[dir][1][lane >> 2], bs2 = ws->bs[dir][2][lane >> 2], bs3 = ws->bs[dir][3][lane >> 2];
                                const int qpq = mq->qp, aoff = mq->alpha_off, boff = mq->beta_off;
                                const int qpe = mn ? (mn->qp + qpq + 1) >> 1 : qpq;
                                const int ia0 = clip3(0, 51, qpe + aoff), ib0 = clip3(0, 51, qpe + boff);
                                const int ia1 = clip3(0, 51, qpq + aoff), ib1 = clip3(0, 51, qpq + boff);
                                const int a0 = sh.alpha[ia0], b0 = sh.beta[ib0], a1 = sh.alpha[ia1], b1 = sh.beta[ib1];
                                filter_edge<4, false>(px, bs0, a0, b0, sh.tc0[ia0][bs0 & 3]);
                                filter_edge<8, false>(px, bs1, a1, b1, sh.tc0[ia1][bs1 & 3]);
                                filter_edge<12, false>(px, bs2, a1, b1, sh.tc0[ia1][bs2 & 3]);
                                filter_edge<16, false>(px, bs3, a1, b1, sh.tc0[ia1][bs3 & 3]);
                                if (dir == 0) {
#pragma unroll
                                    for (int d = 0; d < 5; d++)
                                        *reinterpret_cast<uint32_t *>(&tl->y[4 + lane][d * 4]) =
                                            static_cast<uint32_t>(px[4 * d]) | (px[4 * d + 1] << 8) | (px[4 * d + 2] << 16) | (static_cast<uint32_t>(px[4 * d + 3]) << 24);
                                } else {
#pragma unroll
                                    for (int r = 1; r < 19; r++) tl->y[r][4 + lane] = static_cast<uint8_t>(px[r]);
                                }
                            } else if (lane < 32) { // chroma: plane = bit 3, row/column = low 3 bits; luma edges 0 and 2
                                const int c = (lane >> 3) & 1, i = lane & 7;
                                int px[12];
                                if (dir == 0) {
#pragma unroll
                                    for (int d = 0; d < 3; d++) {
                                        uint32_t w = *reinterpret_cast<const uint32_t *>(&tl->c[c][4 + i][d * 4]);
                                        px[4 * d] = w & 255, px[4 * d + 1] = (w >> 8) & 255, px[4 * d + 2] = (w >> 16) & 255, px[4 * d + 3] = w >> 24;
                                    }
                                } else {
#pragma unroll
                                    for (int r = 0; r < 12; r++) px[r] = tl->c[c][r][4 + i];
                                }
                                const int bs0 = ws->bs[dir][0][i >> 1], bs2 = ws->bs[dir][2][i >> 1];
                                const int qpq = mq->qpc[c], aoff = mq->alpha_off, boff = mq->beta_off;
                                const int qpe = mn ? (mn->qpc[c] + qpq + 1) >> 1 : qpq;
                                const int ia0 = clip3(0, 51, qpe + aoff), ib0 = clip3(0, 51, qpe + boff);
                                const int ia1 = clip3(0, 51, qpq + aoff), ib1 = clip3(0, 51, qpq + boff);
                                filter_edge<4, true>(px, bs0, sh.alpha[ia0], sh.beta[ib0], sh.tc0[ia0][bs0 & 3]);
                                filter_edge<8, true>(px, bs2, sh.alpha[ia1], sh.beta[ib1], sh.tc0[ia1][bs2 & 3]);
                                if (dir == 0) {
#pragma unroll
                                    for (int d = 0; d < 3; d++)
                                        *reinterpret_cast<uint32_t *>(&tl->c[c][4 + i][d * 4]) =
                                            static_cast<uint32_t>(px[4 * d]) | (px[4 * d + 1] << 8) | (px[4 * d + 2] << 16) | (static_cast<uint32_t>(px[4 * d + 3]) << 24);
                                } else {
#pragma unroll
                                    for (int r = 2; r < 10; r++) tl->c[c][r][4 + i] = static_cast<uint8_t>(px[r]);
                                }
                            }
                        }
                        WAVE_SYNC();
                    }
                    // ---- write back rows -3..15, columns -4..15 (the untouched top-left corner is skipped) ----
                    const bool has_left = mbx > 0;
                    for (int i = lane; i < 100; i += 64) {
                        int r = i / 5, d = i - r * 5;
                        if (r >= 1 && (r >= 4 || has_top) && (d >= 1 || has_left) && !(r < 4 && d == 0))
                            *reinterpret_cast<uint32_t *>(Y + static_cast<ptrdiff_t>(r - 4) * W + (d - 1) * 4) = *reinterpret_cast<const uint32_t *>(&tl->y[r][d * 4]);
                    }
                    for (int i = lane; i < 72; i += 64) {
                        int c = i / 36, rem = i - c * 36, r = rem / 3, d = rem - r * 3;
                        if (r >= 3 && (r >= 4 || has_top) && (d >= 1 || has_left) && !(r < 4 && d == 0))
                            *reinterpret_cast<uint32_t *>((c ? C1 : C0) + static_cast<ptrdiff_t>(r - 4) * Wc + (d - 1) * 4) = *reinterpret_cast<const uint32_t *>(&tl->c[c][r][d * 4]);
                    }
                }
            }
            // publish progress: the release orders this wave's global stores before the counter update
            if (lane == 0) __hip_atomic_store(&sh.prog[mby], mbx + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

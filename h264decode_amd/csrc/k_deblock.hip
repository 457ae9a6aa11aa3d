// h264decode_amd/csrc/k_deblock.hip -- K5: in-loop deblocking filter (ITU-T H.264 8.7), gfx950.
//
// 8.7 is specified per macroblock in raster order (vertical edges left to right, then horizontal
// edges top to bottom), and the left-edge filter of MB(x+1,y) rewrites columns of MB(x,y) AFTER
// MB(x,y)'s horizontal edges were filtered, so a whole-picture "all vertical, then all horizontal"
// pass is not bit-exact.  The exact dependency is MB(x,y) after MB(x-1,y) and MB(x+1,y-1): a 2-D
// wavefront.
//
// Mapping.  One workgroup owns a picture, so no cross-CU hand-off is needed; the host picks the number
// of wavefronts (blockDim.x / 64, at most MI_DEBLOCK_MAX_WAVES) so that the groups of 4 macroblock rows
// are covered in as few rounds as possible (1080p: 17 groups -> 9 wavefronts, 2 rounds).  A wavefront owns a GROUP of 4 consecutive macroblock rows and filters four macroblocks
// per step -- 16 lanes each -- staggered along the wavefront diagonal: at step t sub-row k works on
// MB (t - 2k, 4g + k).  All four are independent by construction, so the 64 lanes are busy and the
// per-step latency is shared by 4 macroblocks.
//   * inside a group the 4 sample rows handed from sub-row k-1 to sub-row k travel through a small
//     LDS ring (4 macroblock columns), never through HBM;
//   * between groups (different wavefronts of the workgroup) they travel through a second LDS ring of
//     `ring` macroblock columns per in-flight group (chosen by the host, see mi_deblock_plan), ordered by two LDS counters per group
//     (columns finished by its last row / columns consumed by its first row: back-pressure);
//     no HBM access sits on the dependency path: samples are loaded once (prefetch) and stored once,
//     fire-and-forget -- rows 13..15 of a macroblock are written by the macroblock BELOW it (which
//     modifies them last), so no address is ever stored twice;
//   * MbRecs are prefetched one step ahead, the macroblock's own samples four macroblocks (one 64-byte line
//     per lane) at a time; the 4 columns to the left are carried over from the previous tile in LDS;
//   * a lane filters a whole line of samples in registers (4 luma edges, then 2 chroma edges).
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
struct DbSub { // state of one of the 4 macroblock rows a wavefront works on
    DbTile tile[2];      // double buffer: the left 4 columns of tile[k] come from tile[k^1]
    MbRec rec[3];        // cur / left alternate in [0],[1]; [2] = macroblock above
    uint8_t bs[2][4][4]; // [dir][edge][segment]
    // bottom rows of this sub-row's macroblocks for the sub-row below: ring over 4 MB columns
    uint8_t bot_y[4][4][16];    // [column & 3][row 12..15][x]
    uint8_t bot_c[4][2][2][8];  // [column & 3][plane][row 6..7][x]
    uint8_t pad[32];            // sub-row stride = 8 dwords (mod 32 banks): the 4 sub-rows of a wavefront do not hit the same banks
    // output staging: finished samples of an aligned group of 4 macroblocks, flushed as whole 64-byte (luma) / 32-byte
    // (chroma) lines -- storing 16 bytes per step left every line in HBM as four partial writes (4x write traffic)
    alignas(16) uint8_t ost_y[16][64];
    alignas(16) uint8_t ost_c[2][8][32];
};
struct DbWave {
    DbSub sub[4];
};
struct GroupSlot { // bottom rows of one macroblock column handed to the group below
    uint8_t y[4][16];   // rows 12..15
    uint8_t c[2][2][8]; // [plane][rows 6..7]
};
struct DbShared { // followed in dynamic LDS by DbWave[nwaves] and GroupSlot[nwaves][ring]
    uint8_t alpha[52], beta[52], tc0[52][4];
    int prog[96]; // per group: macroblocks finished in its LAST row
    int cons[96]; // per group: hand-off slots consumed by its FIRST row
};

#if MI_DB_STATS /* diagnostics: shader clocks per phase of wavefront `wave`, written into the pad bytes of MbRec[wave] of the picture */
#define DB_T0() uint64_t db_mark = __builtin_readcyclecounter(); uint32_t db_acc[4] = {0, 0, 0, 0}
#define DB_T(k) do { const uint64_t now_ = __builtin_readcyclecounter(); db_acc[k] += static_cast<uint32_t>((now_ - db_mark) >> 4); db_mark = now_; } while (0)
#else
#define DB_T0() ((void)0)
#define DB_T(k) ((void)0)
#endif
static_assert(sizeof(DbShared) <= MI_DEBLOCK_HDR_BYTES && sizeof(DbWave) == MI_DEBLOCK_WAVE_BYTES && sizeof(GroupSlot) == MI_DEBLOCK_SLOT_BYTES, "LDS layout constants");

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }

// filter one edge of a line of samples held in registers (8.7.2.3 / 8.7.2.4); q0 = px[Q].
// Written without per-lane branches: both filters are evaluated with selects, and the only branches are
// wave-uniform (ballot) skips -- "no lane filters this edge" and "no lane needs the bS 4 filter".
template <int Q, bool CHROMA, int N>
__device__ __forceinline__ void filter_edge(int (&px)[N], int bs, int alpha, int beta, int tc0) {
    const int p0 = px[Q - 1], p1 = px[Q - 2], q0 = px[Q], q1 = px[Q + 1];
    const bool on = bs != 0 && iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta;
    if (__builtin_amdgcn_ballot_w64(on) == 0) return;
    const bool strong = on && bs == 4;
    int np0, nq0;
    if (CHROMA) {
        const int tc = tc0 + 1;
        const int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        np0 = clip3(0, 255, p0 + delta), nq0 = clip3(0, 255, q0 - delta);
        if (__builtin_amdgcn_ballot_w64(strong) != 0) {
            np0 = strong ? (2 * p1 + p0 + q1 + 2) >> 2 : np0;
            nq0 = strong ? (2 * q1 + q0 + p1 + 2) >> 2 : nq0;
        }
        px[Q - 1] = on ? np0 : p0, px[Q] = on ? nq0 : q0;
        return;
    } else {
        const int p2 = px[Q - 3], q2 = px[Q + 2];
        const bool ap = iabs(p2 - p0) < beta, aq = iabs(q2 - q0) < beta;
        const int tc = tc0 + (ap ? 1 : 0) + (aq ? 1 : 0);
        const int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        const int avg = (p0 + q0 + 1) >> 1;
        np0 = clip3(0, 255, p0 + delta), nq0 = clip3(0, 255, q0 - delta);
        int np1 = ap ? p1 + clip3(-tc0, tc0, (p2 + avg - (p1 << 1)) >> 1) : p1;
        int nq1 = aq ? q1 + clip3(-tc0, tc0, (q2 + avg - (q1 << 1)) >> 1) : q1;
        int np2 = p2, nq2 = q2;
        if (__builtin_amdgcn_ballot_w64(strong) != 0) {
            const int p3 = px[Q - 4], q3 = px[Q + 3];
            const bool small = iabs(p0 - q0) < ((alpha >> 2) + 2);
            const bool sp = strong && ap && small, sq = strong && aq && small;
            const int s3 = p0 + q0 + p1 + 2; // shared partial sums of the 4- and 5-tap filters
            np0 = sp ? (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3 : (strong ? (2 * p1 + p0 + q1 + 2) >> 2 : np0);
            np1 = sp ? (p2 + s3) >> 2 : (strong ? p1 : np1);
            np2 = sp ? (2 * p3 + 3 * p2 + s3 + 2) >> 3 : p2;
            const int t3 = p0 + q0 + q1 + 2;
            nq0 = sq ? (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3 : (strong ? (2 * q1 + q0 + p1 + 2) >> 2 : nq0);
            nq1 = sq ? (q2 + t3) >> 2 : (strong ? q1 : nq1);
            nq2 = sq ? (2 * q3 + 3 * q2 + t3 + 2) >> 3 : q2;
        }
        px[Q - 1] = on ? np0 : p0, px[Q] = on ? nq0 : q0;
        px[Q - 2] = on ? np1 : p1, px[Q + 1] = on ? nq1 : q1;
        px[Q - 3] = on ? np2 : p2, px[Q + 2] = on ? nq2 : q2;
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

extern "C" __global__ void __launch_bounds__(MI_DEBLOCK_MAX_WAVES * 64) k_deblock(const uint32_t *pic_list, const PicDesc *pics, const FramePool *pools,
                                                                                  const DevTables *tab, const MbRec *mbrec, int ring) {
    extern __shared__ uint4 dyn_lds[];
    const int nthreads = static_cast<int>(blockDim.x), nwaves = nthreads >> 6;
    DbShared &sh = *reinterpret_cast<DbShared *>(dyn_lds);
    DbWave *waves = reinterpret_cast<DbWave *>(reinterpret_cast<uint8_t *>(dyn_lds) + MI_DEBLOCK_HDR_BYTES);
    GroupSlot *gring = reinterpret_cast<GroupSlot *>(waves + nwaves);
    const int tid = static_cast<int>(threadIdx.x), lane = tid & 63, wave = tid >> 6;
    const int sub = lane >> 4, li = lane & 15; // sub-row inside the group, lane inside the macroblock
    const PicDesc *pd = &pics[pic_list[blockIdx.x]];
    const int wmb = static_cast<int>(pd->wmb), hmb = static_cast<int>(pd->hmb);
    const FramePool *pool = &pools[pd->stream];
    const int W = wmb * 16, H = hmb * 16, Wc = W / 2; // the picture's own geometry
    uint8_t *py = reinterpret_cast<uint8_t *>(pool->base) + static_cast<size_t>(pd->slot) * pool->slot_bytes;
    uint8_t *pcb = py + static_cast<size_t>(W) * H, *pcr = pcb + static_cast<size_t>(W) * H / 4;
    for (int i = tid; i < 96; i += nthreads) sh.prog[i] = 0, sh.cons[i] = 0;
    for (int i = tid; i < 52; i += nthreads) {
        sh.alpha[i] = tab->alpha[i], sh.beta[i] = tab->beta[i];
        sh.tc0[i][0] = 0, sh.tc0[i][1] = tab->tc0[i][1], sh.tc0[i][2] = tab->tc0[i][2], sh.tc0[i][3] = tab->tc0[i][3];
    }
    __syncthreads();
    DbSub *ss = &waves[wave].sub[sub];
    const DbSub *sup = sub > 0 ? &waves[wave].sub[sub - 1] : nullptr; // the sub-row above (same wavefront)
    const MbRec *recs = mbrec + pd->mb_base;
    const int ngroups = (hmb + 3) >> 2;
    DB_T0();
    for (int g = wave; g < ngroups; g += nwaves) {
        const int mby = g * 4 + sub;
        const bool row_ok = mby < hmb, has_top = mby > 0;
        const MbRec *row = recs + static_cast<size_t>(row_ok ? mby : 0) * wmb;
        const int last_sub = min(3, hmb - 1 - g * 4); // last valid sub-row of this group
        // Prefetch registers.  Samples are fetched one aligned group of 4 macroblocks at a time -- the whole 64-byte
        // line of a luma row (32 bytes of a chroma row) by one lane with back-to-back loads -- so that every line
        // leaves HBM once; fetching 16 bytes per step let the line be evicted between its four uses (4x read traffic).
        // MbRecs (cur, top: 2 x 128 bytes as 4 dwords per lane) are whole lines already and stay per step.
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        uint4 gy0 = z4, gy1 = z4, gy2 = z4, gy3 = z4, gc0 = z4, gc1 = z4, pre_rec = z4;
        auto prefetch_samples = [&](int gb) { // gb = first macroblock of the aligned group
            if (!row_ok || gb < 0 || gb >= wmb) return;
            const uint8_t *yrow = py + static_cast<size_t>(mby * 16 + li) * W + gb * 16;
            const uint8_t *crow = (li < 8 ? pcb : pcr) + static_cast<size_t>(mby * 8 + (li & 7)) * Wc + gb * 8;
            const int left = wmb - gb; // macroblocks from gb to the end of the row (>= 1)
            gy0 = *reinterpret_cast<const uint4 *>(yrow);
            if (left > 1) gy1 = *reinterpret_cast<const uint4 *>(yrow + 16);
            if (left > 2) gy2 = *reinterpret_cast<const uint4 *>(yrow + 32);
            if (left > 3) gy3 = *reinterpret_cast<const uint4 *>(yrow + 48);
            if (left > 1)
                gc0 = *reinterpret_cast<const uint4 *>(crow);
            else {
                const uint2 h = *reinterpret_cast<const uint2 *>(crow);
                gc0 = make_uint4(h.x, h.y, 0, 0);
            }
            if (left > 3)
                gc1 = *reinterpret_cast<const uint4 *>(crow + 16);
            else if (left > 2) {
                const uint2 h = *reinterpret_cast<const uint2 *>(crow + 16);
                gc1 = make_uint4(h.x, h.y, 0, 0);
            }
        };
        auto prefetch_rec = [&](int mbx) {
            if (!row_ok || mbx < 0 || mbx >= wmb) return;
            // lanes 0-7: cur record (8 x 16 B), lanes 8-15: record above
            const MbRec *src = li < 8 ? row + mbx : (has_top ? row + mbx - wmb : row + mbx);
            pre_rec = reinterpret_cast<const uint4 *>(src)[li & 7];
        };
        prefetch_samples(0);
        prefetch_rec(-2 * sub); // step 0 (only sub-row 0 is active)
        // this wavefront's ring row was last used by group g - nwaves: its reader (g - nwaves + 1) must be through with it
        if (g >= nwaves && g + 1 < ngroups) {
            while (__hip_atomic_load(&sh.cons[g - nwaves + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < wmb) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
        const int nsteps = wmb + 6;
        for (int t = 0; t < nsteps; t++) {
            const int mbx = t - 2 * sub;
            const bool active = row_ok && mbx >= 0 && mbx < wmb;
            DbTile *tl = &ss->tile[t & 1], *prev = &ss->tile[(t & 1) ^ 1];
            const int cur_slot = t & 1;
            MbRec *mq = &ss->rec[cur_slot], *mleft_rec = &ss->rec[cur_slot ^ 1], *mtop_rec = &ss->rec[2];
            DB_T(3);
            // ---- commit the prefetched data to LDS ----
            if (active) {
                reinterpret_cast<uint4 *>(li < 8 ? mq : mtop_rec)[li & 7] = pre_rec;
                const int k4 = mbx & 3;
                const uint4 pre_y = k4 == 0 ? gy0 : (k4 == 1 ? gy1 : (k4 == 2 ? gy2 : gy3));
                const uint4 pc4 = k4 < 2 ? gc0 : gc1;
                const uint2 pre_c = (k4 & 1) ? make_uint2(pc4.z, pc4.w) : make_uint2(pc4.x, pc4.y);
                uint32_t *yr = reinterpret_cast<uint32_t *>(&tl->y[4 + li][4]);
                yr[0] = pre_y.x, yr[1] = pre_y.y, yr[2] = pre_y.z, yr[3] = pre_y.w;
                uint32_t *cr = reinterpret_cast<uint32_t *>(&tl->c[li >> 3][4 + (li & 7)][4]);
                cr[0] = pre_c.x, cr[1] = pre_c.y;
                if (mbx > 0) { // left 4 columns: carried over from the previous tile
                    *reinterpret_cast<uint32_t *>(&tl->y[4 + li][0]) = *reinterpret_cast<const uint32_t *>(&prev->y[4 + li][16]);
                    *reinterpret_cast<uint32_t *>(&tl->c[li >> 3][4 + (li & 7)][0]) = *reinterpret_cast<const uint32_t *>(&prev->c[li >> 3][4 + (li & 7)][8]);
                }
                // rows above from the sub-row above (same wavefront): LDS ring, final since the previous step
                if (sub > 0) {
                    if (li < 4) {
                        const uint32_t *src = reinterpret_cast<const uint32_t *>(sup->bot_y[mbx & 3][li]);
                        uint32_t *dst = reinterpret_cast<uint32_t *>(&tl->y[li][4]);
                        dst[0] = src[0], dst[1] = src[1], dst[2] = src[2], dst[3] = src[3];
                    } else if (li < 8) {
                        const uint32_t *src = reinterpret_cast<const uint32_t *>(sup->bot_c[mbx & 3][(li >> 1) & 1][li & 1]);
                        uint32_t *dst = reinterpret_cast<uint32_t *>(&tl->c[(li >> 1) & 1][2 + (li & 1)][4]);
                        dst[0] = src[0], dst[1] = src[1];
                    }
                }
            }
            // ---- sub-row 0: rows above come from the previous group (another wavefront) through the group ring ----
            {
                const int x0 = t; // macroblock of sub-row 0 in this step
                if (g > 0 && x0 < wmb) {
                    const int need = min(x0 + 2, wmb);
                    while (__hip_atomic_load(&sh.prog[g - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
                    asm volatile("" ::: "memory");
                    if (sub == 0) {
                        const GroupSlot *gs = &gring[((g - 1) % nwaves) * ring + x0 % ring];
                        if (li < 4) {
                            const uint32_t *src = reinterpret_cast<const uint32_t *>(gs->y[li]);
                            uint32_t *dst = reinterpret_cast<uint32_t *>(&tl->y[li][4]);
                            dst[0] = src[0], dst[1] = src[1], dst[2] = src[2], dst[3] = src[3];
                        } else if (li < 8) {
                            const uint32_t *src = reinterpret_cast<const uint32_t *>(gs->c[(li >> 1) & 1][li & 1]);
                            uint32_t *dst = reinterpret_cast<uint32_t *>(&tl->c[(li >> 1) & 1][2 + (li & 1)][4]);
                            dst[0] = src[0], dst[1] = src[1];
                        }
                    }
                }
                // back-pressure: the ring slot this step's last sub-row will overwrite held column xl - RING of this
                // group; the group below must have consumed it
                const int xl = t - 2 * last_sub;
                if (g + 1 < ngroups && xl >= ring && xl < wmb) {
                    while (__hip_atomic_load(&sh.cons[g + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < xl - ring + 1) __builtin_amdgcn_s_sleep(1);
                    asm volatile("" ::: "memory");
                }
            }
            DB_T(0);
            WAVE_SYNC();
            if (g > 0 && t < wmb) { // the hand-off slot of column t has been copied into the tile
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(&sh.cons[g], t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if ((mbx & 3) == 3) prefetch_samples(mbx + 1); // the group's registers are free: fetch the next one (used from the next step on)
            prefetch_rec(mbx + 1);
            // ---- boundary strengths: 32 per macroblock, 2 per lane ----
            const MbRec *ml = nullptr, *mt = nullptr;
            int dbf = 1;
            if (active) {
                dbf = mq->dbf_idc;
                ml = mbx > 0 ? mleft_rec : nullptr, mt = has_top ? mtop_rec : nullptr;
                if (dbf == 2) { // no filtering across slice boundaries
                    if (ml && ml->slice_in_pic != mq->slice_in_pic) ml = nullptr;
                    if (mt && mt->slice_in_pic != mq->slice_in_pic) mt = nullptr;
                }
                if (dbf != 1) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int idx = li + 16 * h, dir = idx >> 4, e = (idx >> 2) & 3, k = idx & 3;
                        const MbRec *mn = dir == 0 ? ml : mt;
                        int bs = 0;
                        if (!(e == 0 && !mn) && !((e & 1) && mq->t8x8)) {
                            const MbRec *mp = e == 0 ? mn : mq;
                            int qb = dir == 0 ? k * 4 + e : e * 4 + k;
                            int pb = dir == 0 ? k * 4 + (e == 0 ? 3 : e - 1) : (e == 0 ? 3 : e - 1) * 4 + k;
                            bs = edge_bs(mp, pb, mq, qb, e == 0);
                        }
                        ss->bs[dir][e][k] = static_cast<uint8_t>(bs);
                    }
                }
            }
            WAVE_SYNC();
            DB_T(1);
            // ---- the two filtering passes: a whole line of samples in registers per lane ----
            const bool filt = active && dbf != 1;
            for (int dir = 0; dir < 2; dir++) {
                if (filt) {
                    const MbRec *mn = dir == 0 ? ml : mt;
                    const uint32_t b0w = *reinterpret_cast<const uint32_t *>(ss->bs[dir][0]), b1w = *reinterpret_cast<const uint32_t *>(ss->bs[dir][1]);
                    const uint32_t b2w = *reinterpret_cast<const uint32_t *>(ss->bs[dir][2]), b3w = *reinterpret_cast<const uint32_t *>(ss->bs[dir][3]);
                    const int aoff = mq->alpha_off, boff = mq->beta_off;
                    if (b0w | b1w | b2w | b3w) {
                        { // luma: lane li = row (dir 0) or column (dir 1)
                            int px[20];
                            if (dir == 0) {
#pragma unroll
                                for (int d = 0; d < 5; d++) {
                                    uint32_t w = *reinterpret_cast<const uint32_t *>(&tl->y[4 + li][d * 4]);
                                    px[4 * d] = w & 255, px[4 * d + 1] = (w >> 8) & 255, px[4 * d + 2] = (w >> 16) & 255, px[4 * d + 3] = w >> 24;
                                }
                            } else {
#pragma unroll
                                for (int r = 0; r < 20; r++) px[r] = tl->y[r][4 + li];
                            }
                            const int sh8 = 8 * (li >> 2);
                            const int bs0 = (b0w >> sh8) & 255, bs1 = (b1w >> sh8) & 255, bs2 = (b2w >> sh8) & 255, bs3 = (b3w >> sh8) & 255;
                            const int qpq = mq->qp;
                            const int qpe = mn ? (mn->qp + qpq + 1) >> 1 : qpq;
                            const int ia0 = clip3(0, 51, qpe + aoff), ib0 = clip3(0, 51, qpe + boff);
                            const int ia1 = clip3(0, 51, qpq + aoff), ib1 = clip3(0, 51, qpq + boff);
                            const int a0 = sh.alpha[ia0], be0 = sh.beta[ib0], a1 = sh.alpha[ia1], be1 = sh.beta[ib1];
                            filter_edge<4, false>(px, bs0, a0, be0, sh.tc0[ia0][bs0 & 3]);
                            filter_edge<8, false>(px, bs1, a1, be1, sh.tc0[ia1][bs1 & 3]);
                            filter_edge<12, false>(px, bs2, a1, be1, sh.tc0[ia1][bs2 & 3]);
                            filter_edge<16, false>(px, bs3, a1, be1, sh.tc0[ia1][bs3 & 3]);
                            if (dir == 0) {
#pragma unroll
                                for (int d = 0; d < 5; d++)
                                    *reinterpret_cast<uint32_t *>(&tl->y[4 + li][d * 4]) =
                                        static_cast<uint32_t>(px[4 * d]) | (px[4 * d + 1] << 8) | (px[4 * d + 2] << 16) | (static_cast<uint32_t>(px[4 * d + 3]) << 24);
                            } else {
#pragma unroll
                                for (int r = 1; r < 19; r++) tl->y[r][4 + li] = static_cast<uint8_t>(px[r]);
                            }
                        }
                        { // chroma: plane = li >> 3, row/column = li & 7; luma edges 0 and 2
                            const int c = li >> 3, i = li & 7;
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
                            const int sh8 = 8 * (i >> 1);
                            const int bs0 = (b0w >> sh8) & 255, bs2 = (b2w >> sh8) & 255;
                            const int qpq = mq->qpc[c];
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
                }
                WAVE_SYNC();
            }
            DB_T(2);
            // ---- results.  Own rows 0..12 (0..15 in the last picture row) go to the staging group and reach HBM as whole
            // lines once the group is complete; rows -3..-1 of the macroblock above (this macroblock modified them last) are
            // stored directly.  LDS rings: bottom rows for the sub-row / group below ----
            if (active) {
                uint8_t *Y = py + static_cast<size_t>(mby * 16) * W + mbx * 16;
                const bool has_left = mbx > 0, last_row = mby == hmb - 1, row_end = mbx == wmb - 1;
                const int k4 = mbx & 3;
                const uint32_t *r = reinterpret_cast<const uint32_t *>(&tl->y[4 + li][0]);
                const uint32_t *cr = reinterpret_cast<const uint32_t *>(&tl->c[li >> 3][4 + (li & 7)][0]);
                uint32_t *oy = reinterpret_cast<uint32_t *>(ss->ost_y[li]);
                uint32_t *oc = reinterpret_cast<uint32_t *>(ss->ost_c[li >> 3][li & 7]);
                const bool store_y = li < 13 || last_row, store_c = (li & 7) < 7 || last_row;
                // a lane stages and flushes only its own rows, and LDS operations of a wavefront complete in order: no sync needed
                auto flush = [&](int first_mb, int n_mb) { // macroblocks first_mb .. first_mb + n_mb - 1 of this row, n_mb = 1..4
                    WAVE_SYNC(); // order the dword stores into the staging rows before the wide reads below
                    uint8_t *Yd = py + static_cast<size_t>(mby * 16 + li) * W + first_mb * 16;
                    uint8_t *Cd = (li < 8 ? pcb : pcr) + static_cast<size_t>(mby * 8 + (li & 7)) * Wc + first_mb * 8;
                    const uint4 *sy = reinterpret_cast<const uint4 *>(oy);
                    const uint2 *sc = reinterpret_cast<const uint2 *>(oc);
                    if (store_y) {
                        if (n_mb == 4) {
                            const uint4 a = sy[0], b = sy[1], c = sy[2], d = sy[3];
                            uint4 *dst = reinterpret_cast<uint4 *>(Yd);
                            dst[0] = a, dst[1] = b, dst[2] = c, dst[3] = d;
                        } else
                            for (int k = 0; k < n_mb; k++) reinterpret_cast<uint4 *>(Yd)[k] = sy[k];
                    }
                    if (store_c) {
                        if (n_mb == 4) {
                            const uint4 a = reinterpret_cast<const uint4 *>(oc)[0], b = reinterpret_cast<const uint4 *>(oc)[1];
                            reinterpret_cast<uint4 *>(Cd)[0] = a, reinterpret_cast<uint4 *>(Cd)[1] = b;
                        } else
                            for (int k = 0; k < n_mb; k++) reinterpret_cast<uint2 *>(Cd)[k] = sc[k];
                    }
                    WAVE_SYNC(); // ... and the reads before the next group's stores
                };
                // columns 12..15 (chroma 4..7) of the macroblock to the left are final now
                if (has_left) {
                    const int kl = (k4 + 3) & 3; // its slot in the staging group
                    oy[kl * 4 + 3] = r[0];
                    oc[kl * 2 + 1] = cr[0];
                    if (k4 == 0) flush(mbx - 4, 4); // that completed the previous group
                }
                oy[k4 * 4 + 0] = r[1], oy[k4 * 4 + 1] = r[2], oy[k4 * 4 + 2] = r[3];
                oc[k4 * 2] = cr[1];
                if (row_end) { // no macroblock to the right: the last columns are final too
                    oy[k4 * 4 + 3] = r[4];
                    oc[k4 * 2 + 1] = cr[2];
                    flush(mbx - k4, k4 + 1);
                }
                if (has_top) { // rows -3..-1 (luma), -1 (chroma), columns 0..15 / 0..7: always, the macroblock above never stores them
                    if (li < 3) {
                        const uint32_t *r = reinterpret_cast<const uint32_t *>(&tl->y[1 + li][4]);
                        *reinterpret_cast<uint4 *>(Y + static_cast<ptrdiff_t>(li - 3) * W) = make_uint4(r[0], r[1], r[2], r[3]);
                    } else if (li >= 8 && li < 10) {
                        const int c = li - 8;
                        const uint32_t *r = reinterpret_cast<const uint32_t *>(&tl->c[c][3][4]);
                        *reinterpret_cast<uint2 *>((c ? pcr : pcb) + static_cast<size_t>(mby * 8 - 1) * Wc + mbx * 8) = make_uint2(r[0], r[1]);
                    }
                }
                // rings: own bottom rows (columns 12..15 still provisional) and the now final columns 12..15 of the left MB
                if (!last_row && li < 8) {
                    uint32_t *dy, *dyl = nullptr, *dc, *dcl = nullptr;
                    const int c = (li >> 1) & 1, r = li & 1;
                    if (sub < last_sub) {
                        dy = reinterpret_cast<uint32_t *>(ss->bot_y[mbx & 3][li & 3]), dc = reinterpret_cast<uint32_t *>(ss->bot_c[mbx & 3][c][r]);
                        if (has_left) dyl = reinterpret_cast<uint32_t *>(&ss->bot_y[(mbx - 1) & 3][li & 3][12]), dcl = reinterpret_cast<uint32_t *>(&ss->bot_c[(mbx - 1) & 3][c][r][4]);
                    } else {
                        GroupSlot *row = &gring[(g % nwaves) * ring];
                        GroupSlot *gs = &row[mbx % ring], *gl = &row[(mbx + ring - 1) % ring];
                        dy = reinterpret_cast<uint32_t *>(gs->y[li & 3]), dc = reinterpret_cast<uint32_t *>(gs->c[c][r]);
                        if (has_left) dyl = reinterpret_cast<uint32_t *>(&gl->y[li & 3][12]), dcl = reinterpret_cast<uint32_t *>(&gl->c[c][r][4]);
                    }
                    if (li < 4) {
                        const uint32_t *src = reinterpret_cast<const uint32_t *>(&tl->y[16 + li][0]);
                        dy[0] = src[1], dy[1] = src[2], dy[2] = src[3], dy[3] = src[4];
                        if (dyl) dyl[0] = src[0];
                    } else {
                        const uint32_t *src = reinterpret_cast<const uint32_t *>(&tl->c[c][10 + r][0]);
                        dc[0] = src[1], dc[1] = src[2];
                        if (dcl) dcl[0] = src[0];
                    }
                }
            }
            WAVE_SYNC();
            // progress of the group's last row (LDS-only hand-off: LDS operations of a wavefront complete in order)
            {
                const int xl = t - 2 * last_sub;
                if (xl >= 0 && xl < wmb) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&sh.prog[g], xl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
#if MI_DB_STATS
    if (lane == 0) {
        uint32_t *dst = reinterpret_cast<uint32_t *>(const_cast<MbRec *>(recs + wave)->pad);
        dst[0] = db_acc[0], dst[1] = db_acc[1], dst[2] = db_acc[2], dst[3] = db_acc[3];
    }
#endif
}

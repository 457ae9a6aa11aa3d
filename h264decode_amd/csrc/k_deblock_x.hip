// h264decode_amd/csrc/k_deblock_x.hip -- K5 for launches with fewer pictures than the chip has CUs: in-loop deblocking (ITU-T H.264 8.7) with a
// picture spread over several workgroups, gfx950.  (k_deblock.hip is K5 proper: one workgroup per picture, two lines per lane in packed
// 16-bit arithmetic; this kernel keeps round 2's one-sample-per-lane form -- a launch that needs bands has few pictures, the chain's
// latency is everything and its instruction count per wavefront is what the banded form was tuned for.)
//
// 8.7 is specified per macroblock in raster order (vertical edges left to right, then horizontal
// edges top to bottom), and the left-edge filter of MB(x+1,y) rewrites columns 13..15 of MB(x,y) AFTER
// MB(x,y)'s horizontal edges were filtered, so a whole-picture "all vertical, then all horizontal"
// pass is not bit-exact.  Per 4x4 block the order is: left edge, right edge, top edge, bottom edge --
// except in the last block column of a macroblock, whose right edge belongs to the next macroblock.
// That exception makes every macroblock row one serial chain (V0..V3 of MB x, its horizontal edges in
// columns 12..15, V0 of MB x+1, ...), and the top edge of MB(x,y) needs rows 13..15 of MB(x,y-1) after
// V0 of MB(x+1,y-1).  So a picture is a 2-D wavefront in which row y can trail row y-1 by ONE macroblock,
// provided the vertical-edge pass of a step runs before the horizontal-edge pass of the same step.
//
// Mapping.  A picture is cut into "bands" of consecutive row groups, one workgroup per band, one wavefront per GROUP of 4 consecutive
// macroblock rows ("sub-rows", 16 lanes each), one round.  At step t sub-row k of a group works on macroblock column t - k, so a group
// trails the one above it by 4 steps.  A step is:
//   1. the macroblock's DbPrm (k_dbprep.hip) -> LDS -> this lane's strengths and filter parameters;
//   2. vertical edges: a lane filters one whole line of 20 samples in registers -- 16 fresh from the
//      prefetch registers, 4 (columns 12..15 of the macroblock to the left) from the LDS tile;
//   3. hand-off: those 4 columns are final now, which completes rows 12..15 of the macroblock to the left for the
//      sub-row below (same wavefront: an LDS buffer; next group: an LDS ring ordered by two counters with
//      workgroup-scope release / acquire); then every sub-row picks up the rows above its own macroblock;
//   4. horizontal edges: a lane filters one column of 20 samples;
//   5. finished samples go to HBM: a lane loads the 16 bytes of its row for the NEXT macroblock one step ahead and stores what a step
//      finishes straight from the tile (HBM traffic is no concern at these picture counts, the chain's instruction count is).
// Inside a band: LDS rings, workgroup-scope counters.  Between bands the bottom rows travel
// through a ring in global memory as 8-byte {epoch, data} granules written by ONE agent-scope (sc1) store each and read
// by agent-scope loads until every tag shows this launch's epoch: the data is its own flag, so no fence, no separate
// flag and no assumption about which CU or XCD a band runs on.  A band only ever waits for the band above it, and bands
// take their (picture, band) from a ticket counter in that order, so whoever a workgroup waits for is already running.
// A group can also be worked on by TWO wavefronts, one filtering luma and one chroma (`roles` = 2): the two
// planes share nothing but the boundary strengths, which both wavefronts read from the macroblock's DbPrm, so each is an
// instruction stream about two thirds / one third as long -- and a lone wavefront's step time is its instruction count.
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

struct DbSub { // state of one of the 4 macroblock rows a wavefront works on
    // luma tile, rows -4..15 of the macroblock: bytes 12..15 of a row = columns -4..-1, bytes 16..31 = columns 0..15
    alignas(16) uint8_t y[20][32];
    // chroma tiles, rows -4..7 (-2.. used): bytes 4..7 = columns -4..-1, bytes 8..15 = columns 0..7
    alignas(16) uint8_t c[2][12][16];
    DbPrm prm;           // the current macroblock's strengths and filter parameters (k_dbprep)
    // rows 12..15 (chroma 6..7) of the macroblock this sub-row finished in the previous step, for the sub-row below
    alignas(16) uint8_t bot_y[4][16];
    alignas(8) uint8_t bot_c[2][2][8];
};
struct DbWave {
    DbSub sub[4];
};
struct GroupSlot { // rows 12..15 of one macroblock column handed to the group below
    alignas(16) uint8_t y[4][16];
    alignas(8) uint8_t c[2][2][8];
};
struct DbShared { // followed in dynamic LDS by DbWave[nwaves] and the hand-off rings
    int prog[192]; // per group (x role): macroblock columns of its LAST row that are final (rows 12..15 complete)
    int cons[192]; // per group (x role): hand-off slots consumed by its FIRST row
    uint32_t ticket; // banded builds: which (picture, band) this workgroup drew
};
static_assert(sizeof(DbShared) <= MI_DEBLOCK_HDR_BYTES && sizeof(DbWave) == MI_DEBLOCK_WAVE_BYTES && sizeof(GroupSlot) == MI_DEBLOCK_SLOT_BYTES,
              "LDS layout constants");

// Global memory through an explicit address-space-1 pointer with a wave-uniform base and a 32-bit per-lane offset: the
// frame pointers are built from integers (FramePool::base), which the compiler would otherwise treat as generic (flat_*
// instructions, two wait counters) and keep as 64-bit per-lane pointers in registers for the whole kernel.
typedef __attribute__((address_space(1))) uint8_t g8;
typedef uint32_t v4u __attribute__((ext_vector_type(4))); // native vectors: assignable across address spaces (HIP's uint4 is a struct)
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) v4u g_uint4;
typedef __attribute__((address_space(1))) v2u g_uint2;
#define GLD16(base, off) (*reinterpret_cast<const g_uint4 *>((base) + (off)))
#define GLD8(base, off) (*reinterpret_cast<const g_uint2 *>((base) + (off)))
// (Streaming / non-temporal stores and DbPrm loads were tried in round 4 to keep the XCD's L2 for the sample lines: no faster, and WRITE_SIZE grew
// from 0.85 to 1.10 GB per launch -- partial lines leave the L2 before the rest of the line arrives.)
#define GST16(base, off, v) (*reinterpret_cast<g_uint4 *>((base) + (off)) = (v))
#define GST8(base, off, v) (*reinterpret_cast<g_uint2 *>((base) + (off)) = (v))
typedef __attribute__((address_space(1))) uint32_t g_uint1;
#define GST4(base, off, v) (*reinterpret_cast<g_uint1 *>((base) + (off)) = (v))
// keeps lane-dependent values from being hoisted out of the step loop (dozens of loop-invariant addresses would otherwise
// live in registers for the whole kernel)
#define OPAQUE(x) asm volatile("" : "+v"(x))
// diagnostic build (-DMI_DB_STATS): shader clocks per phase of the step loop, summed over one wavefront's steps, added to xstatus[8 + phase]
// by the wavefront of group 0 of every picture (tools/deblock_phase_probe.py); k_deblock gets the status words as an extra argument in that build
#if defined(MI_DB_STATS)
#define STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += static_cast<uint32_t>(now_ - st_last); st_last = now_; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ int adiff(int a, int b) { // |a - b| for operands in 0..65535 (one v_sad_u16)
    return static_cast<int>(__builtin_amdgcn_sad_u16(static_cast<uint32_t>(a), static_cast<uint32_t>(b), 0u));
}
__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }

// filter one edge of a line of samples held in registers (8.7.2.3 / 8.7.2.4); q0 = px[Q].
// Written without per-lane branches: both filters are evaluated with selects, and the only branches are
// wave-uniform (ballot) skips -- "no lane filters this edge" and "no lane needs the bS 4 filter".
template <int Q, bool CHROMA, int N>
__device__ __forceinline__ void filter_edge(int (&px)[N], int bs, int alpha, int beta, int tc0) {
    const int p0 = px[Q - 1], p1 = px[Q - 2], q0 = px[Q], q1 = px[Q + 1];
    const bool on = bs != 0 && adiff(p0, q0) < alpha && adiff(p1, p0) < beta && adiff(q1, q0) < beta;
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
        const bool ap = adiff(p2, p0) < beta, aq = adiff(q2, q0) < beta;
        const int tc = tc0 + (ap ? 1 : 0) + (aq ? 1 : 0);
        const int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        const int avg = (p0 + q0 + 1) >> 1;
        np0 = clip3(0, 255, p0 + delta), nq0 = clip3(0, 255, q0 - delta);
        int np1 = ap ? p1 + clip3(-tc0, tc0, (p2 + avg - (p1 << 1)) >> 1) : p1;
        int nq1 = aq ? q1 + clip3(-tc0, tc0, (q2 + avg - (q1 << 1)) >> 1) : q1;
        int np2 = p2, nq2 = q2;
        if (__builtin_amdgcn_ballot_w64(strong) != 0) {
            const int p3 = px[Q - 4], q3 = px[Q + 3];
            const bool small = adiff(p0, q0) < ((alpha >> 2) + 2);
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

__device__ __forceinline__ void unpack4(uint32_t w, int &a, int &b, int &c, int &d) {
    a = static_cast<int>(w & 255u), b = static_cast<int>(__builtin_amdgcn_ubfe(w, 8, 8)), c = static_cast<int>(__builtin_amdgcn_ubfe(w, 16, 8)), d = static_cast<int>(w >> 24);
}
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
    return static_cast<uint32_t>(a) | (static_cast<uint32_t>(b) << 8) | (static_cast<uint32_t>(c) << 16) | (static_cast<uint32_t>(d) << 24);
}

typedef __attribute__((address_space(1))) unsigned long long gu64;
extern "C" __global__ void __launch_bounds__(MI_DEBLOCK_MAX_WAVES * 64) k_deblock_x(const uint32_t *pic_list, const PicDesc *pics, const DbPrm *dbprm, int ring, int ring_last,
                                                                                    int last_bufs, unsigned long long *xring_, uint32_t epoch, int nbands, uint32_t *ticket,
                                                                                    uint32_t ticket_base, int wmb_max, uint32_t *xstatus, int roles) {
    extern __shared__ uint4 dyn_lds[];
    const int nthreads = static_cast<int>(blockDim.x), nwaves = nthreads >> 6;
    DbShared &sh = *reinterpret_cast<DbShared *>(dyn_lds);
    DbWave *waves = reinterpret_cast<DbWave *>(reinterpret_cast<uint8_t *>(dyn_lds) + MI_DEBLOCK_HDR_BYTES);
    GroupSlot *rings = reinterpret_cast<GroupSlot *>(waves + nwaves); // region r (written by the groups of wavefront r) starts at r * ring
    const int tid = static_cast<int>(threadIdx.x), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane_v = tid & 63;
    if (tid == 0) sh.ticket = atomicAdd(ticket, 1u) - ticket_base; // tickets go out in (picture, band) order: see the header
    __syncthreads();
    const uint32_t tk = sh.ticket;
    const uint32_t pic_i = tk / static_cast<uint32_t>(nbands);
    const int band = static_cast<int>(tk - pic_i * static_cast<uint32_t>(nbands));
    const PicDesc *pd = &pics[pic_list[pic_i]];
    const int wmb = static_cast<int>(pd->wmb), hmb = static_cast<int>(pd->hmb);
    // the picture's place in its frame slot (PicDesc): W = bytes from one luma row of the PICTURE to the next (a field picture lives in the
    // rows of its parity: twice the frame's pitch, first row y_off bytes in); offsets are relative to the slot's first byte
    const int W = static_cast<int>(pd->pitch), Wc = W / 2;
    g8 *const py = (g8 *)(pd->pool_base + static_cast<uint64_t>(pd->slot) * pd->slot_bytes);
    const uint32_t y_off = pd->field == 2 ? pd->pitch >> 1 : 0u;
    const uint32_t cb_off = pd->plane + (y_off >> 1), cr_off = cb_off + (pd->plane >> 2);
    for (int i = tid; i < 192; i += nthreads) sh.prog[i] = 0, sh.cons[i] = 0;
    __syncthreads();
    const DbPrm *prms = dbprm + pd->mb_base;
    const int ngroups = (hmb + 3) >> 2;
    const v4u z4 = v4u{0u, 0u, 0u, 0u};
    const v2u z2 = v2u{0u, 0u};
    // band b owns the groups [b * ngroups / nbands, (b + 1) * ngroups / nbands), one wavefront each (the host launches
    // enough wavefronts for the largest band); rings: region w of LDS is written by wavefront w, also by the band's last one
    const int g0 = band * ngroups / nbands, g1 = (band + 1) * ngroups / nbands;
    GroupSlot *const in_stage = rings + nwaves * ring; // the slot of the band above, copied from the global ring (the two roles write disjoint parts of it)
    int pband = band - 1;                               // the band that owns group g0 - 1 (bands of small pictures can be empty)
    while (pband > 0 && pband * ngroups / nbands == (pband + 1) * ngroups / nbands) pband--;
    gu64 *const xin = (gu64 *)xring_ + (static_cast<size_t>(pic_i) * nbands + (pband > 0 ? pband : 0)) * static_cast<size_t>(wmb_max) * 24;
    gu64 *const xout = (gu64 *)xring_ + (static_cast<size_t>(pic_i) * nbands + band) * static_cast<size_t>(wmb_max) * 24;
    // roles == 2: wavefront 2k filters the luma of the band's k-th group, wavefront 2k + 1 its chroma
    const int role = roles == 2 ? (wave & 1) : -1, gw = roles == 2 ? wave >> 1 : wave;
    const bool do_l = role != 1, do_c = role != 0;
    for (int g = g0 + gw; g < g1; g += ngroups) { // at most one iteration
        int lane = lane_v;
        OPAQUE(lane);
        const int sub = lane >> 4, li = lane & 15; // sub-row inside the group, lane inside the macroblock
        const int mby = g * 4 + sub;
        const bool row_ok = mby < hmb, has_top = mby > 0, last_row = mby == hmb - 1;
        const int last_sub = min(3, hmb - 1 - g * 4); // last valid sub-row of this group
        const bool feeds_group = g + 1 < ngroups;     // this group's last row hands its bottom rows to group g + 1
        // hand-off rings: the one this group writes (region `wave`) and the one it reads (written by group g - 1)
        // (the last wavefront's region holds whole rows and, from three rounds on, one buffer per round parity: see mi_deblock_plan)
        const bool band_first = gw == 0 && g > 0;              // the rows above come from another workgroup
        const int pc = roles == 2 ? 2 * g + role : g, pc_up = roles == 2 ? 2 * (g - 1) + role : g - 1, pc_dn = roles == 2 ? 2 * (g + 1) + role : g + 1; // counters of this / the upper / the lower group
        const bool to_global = feeds_group && g == g1 - 1;     // the bottom rows go to another workgroup
        const int out_depth = ring;
        GroupSlot *out_ring = rings + wave * ring;
        const int in_depth = band_first ? 1 : ring;
        const GroupSlot *in_ring = band_first ? in_stage : rings + (gw > 0 ? wave - (roles == 2 ? 2 : 1) : 0) * ring;
        // this lane's granule of the slot of column 0 (lanes 0..23: the 24 dwords of a GroupSlot), re-read until its tag matches
        unsigned long long pf = 0;
        const bool gran = lane < 24 && (lane < 16 ? do_l : do_c); // the granules of this wavefront's planes: 16 luma dwords, 8 chroma dwords
        if (band_first && gran) pf = __hip_atomic_load(xin + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t yrow0 = y_off + static_cast<uint32_t>((row_ok ? mby : 0) * 16 + li) * W;                           // this lane's luma row
        const uint32_t crow0 = (li < 8 ? cb_off : cr_off) + static_cast<uint32_t>((row_ok ? mby : 0) * 8 + (li & 7)) * Wc; // and chroma row
        // Banded build: a launch that needs bands has few pictures, so HBM traffic is no concern and the chain's instruction count is
        // everything -- a lane loads the 16 bytes of its row for the NEXT macroblock one step ahead and stores what a step
        // finishes straight from the tile (the four columns the vertical pass just completed + this macroblock's first
        // twelve), instead of moving whole 64-byte lines through four register slots (K5 proper, below): every line then moves four
        // times, the slot bookkeeping -- a fifth of a step -- is gone.
        v4u pre_y = z4;
        v2u pre_c = z2;
        v4u pre_rec = z4;
        auto prefetch_mb = [&](int mbx) { // the lane's row of macroblock column mbx (of this lane's sub-row)
            if (!row_ok || mbx < 0 || mbx >= wmb) return;
            if (do_l) pre_y = GLD16(py, yrow0 + mbx * 16);
            if (do_c) pre_c = GLD8(py, crow0 + mbx * 8);
        };
        auto prefetch_rec = [&](int mbx) { // lanes 0..4 of a sub-row: the five 16-byte pieces of the macroblock's DbPrm
            if (!row_ok || mbx < 0 || mbx >= wmb || li >= 5) return;
            pre_rec = reinterpret_cast<const v4u *>(prms + static_cast<uint32_t>(mby * wmb + mbx))[li];
        };
        const bool up_lane = li >= 13 && !last_row;        // luma: this lane stores a row of the macroblock above
        const bool upc_lane = (li & 7) == 7 && !last_row;  // chroma: rows 7 store row -1 of the macroblock above
        const bool y_stores = !up_lane || has_top, c_stores = !upc_lane || has_top; // up lanes of the first picture row have nothing above
        const uint32_t yout = up_lane && has_top ? yrow0 - 16u * W : yrow0; // row li of the MB above = row li - 16
        const uint32_t cout = upc_lane && has_top ? crow0 - 8u * Wc : crow0;
        prefetch_mb(-sub);
        prefetch_rec(-sub); // step 0 (only sub-row 0 is active)
        // the ring this group writes was last used by the group `reuse` groups earlier: that group's reader must be through with it
        const int nsteps = wmb + 3;
#if defined(MI_DB_STATS)
        uint32_t st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
        for (int t = 0; t < nsteps; t++) {
            STAMP(5); // loop control + whatever the compiler moved across the step boundary
            int lane = lane_v;
            OPAQUE(lane);
            const int sub = lane >> 4, li = lane & 15;
            DbSub *ss = &waves[wave].sub[sub];
            const DbSub *sup = &waves[wave].sub[sub > 0 ? sub - 1 : 0]; // the sub-row above (same wavefront)
            const int mby = g * 4 + sub;
            const bool row_ok = mby < hmb, has_top = mby > 0, last_row = mby == hmb - 1;
            const bool up_lane = li >= 13 && !last_row, upc_lane = (li & 7) == 7 && !last_row;
            const int mbx = t - sub;
            const bool active = row_ok && mbx >= 0 && mbx < wmb;
            // this step's input registers (wave-uniform slot)
            const v4u in_y = pre_y;
            const v2u in_c = pre_c;
            // ---- 1. the macroblock's DbPrm -> LDS -> this lane's strengths and filter parameters ----
            if (active && li < 5) reinterpret_cast<v4u *>(&ss->prm)[li] = pre_rec;
            WAVE_SYNC();
            STAMP(6);
            prefetch_mb(mbx + 1);
            prefetch_rec(mbx + 1);
            STAMP(7);
            // P*[0] luma, [1] chroma (plane li >> 3): bs = the strengths of this lane's segment of edges 0..3 (edge e in byte e; chroma:
            // luma edges 0 and 2), ab = alpha(e0) | beta(e0) << 8 | alpha(inner) << 16 | beta(inner) << 24, tc = tC0 per edge.
            // V: vertical edges, H: horizontal edges.  No table, no record, no division of labour: four 16-byte LDS reads.
            uint32_t Vbs[2] = {0, 0}, Vab[2] = {0, 0}, Vtc[2] = {0, 0}, Hbs[2] = {0, 0}, Hab[2] = {0, 0}, Htc[2] = {0, 0};
            if (active) {
                // DbPrm::bs[segment][direction][edge]: a lane's strengths of the four edges crossing its line are one dword
                const v4u z4p = v4u{0u, 0u, 0u, 0u};
                const v4u blk_l = do_l ? *reinterpret_cast<const v4u *>(&ss->prm.pl[0]) : z4p, blk_c = do_c ? *reinterpret_cast<const v4u *>(&ss->prm.pl[1 + (li >> 3)]) : z4p;
                // a plane's block: a0V b0V a1 b1 | a0H b0H t00 t01 | t02 t10 t11 t12 | t20 t21 t22 pad  (tKb: tC0 of edge kind K -- left /
                // inner / top -- for bS b + 1).  Rows shifted up by one byte, so that bS 0 (and 4: & 3) selects a zero byte.
                auto params = [](v4u blk, uint32_t vbs, uint32_t hbs, uint32_t &vab, uint32_t &vtc, uint32_t &hab, uint32_t &htc) {
                    vab = blk.x;
                    hab = (blk.y & 0xFFFFu) | (blk.x & 0xFFFF0000u);
                    const uint32_t tw0 = ((blk.y >> 16) | ((blk.z & 255u) << 16)) << 8, tw1 = blk.z & 0xFFFFFF00u, tw2 = blk.w << 8;
                    auto sel = [](uint32_t tw, uint32_t bs) { return (tw >> (8u * (bs & 3u))) & 255u; };
                    vtc = sel(tw0, vbs) | (sel(tw1, vbs >> 8) << 8) | (sel(tw1, vbs >> 16) << 16) | (sel(tw1, vbs >> 24) << 24);
                    htc = sel(tw2, hbs) | (sel(tw1, hbs >> 8) << 8) | (sel(tw1, hbs >> 16) << 16) | (sel(tw1, hbs >> 24) << 24);
                };
                if (do_l) {
                    Vbs[0] = *reinterpret_cast<const uint32_t *>(ss->prm.bs[li >> 2][0]), Hbs[0] = *reinterpret_cast<const uint32_t *>(ss->prm.bs[li >> 2][1]);
                    params(blk_l, Vbs[0], Hbs[0], Vab[0], Vtc[0], Hab[0], Htc[0]);
                }
                if (do_c) {
                    Vbs[1] = *reinterpret_cast<const uint32_t *>(ss->prm.bs[(li & 7) >> 1][0]) & 0x00FF00FFu, Hbs[1] = *reinterpret_cast<const uint32_t *>(ss->prm.bs[(li & 7) >> 1][1]) & 0x00FF00FFu; // luma edges 0 and 2
                    params(blk_c, Vbs[1], Hbs[1], Vab[1], Vtc[1], Hab[1], Htc[1]);
                }
            }
            STAMP(0);
            // ---- 2. vertical edges: lane li = luma row li, then chroma (plane li >> 3, row li & 7) ----
            {
                const bool any = (Vbs[0] | Vbs[1]) != 0;
                uint32_t w0 = 0, w1 = in_y.x, w2 = in_y.y, w3 = in_y.z, w4 = in_y.w; // w0 = columns -4..-1
                uint32_t c0 = 0, c1 = in_c.x, c2 = in_c.y;
                if (active && mbx > 0) {
                    if (do_l) w0 = *reinterpret_cast<const uint32_t *>(&ss->y[4 + li][28]); // columns 12..15 of the previous macroblock, after its horizontal pass
                    if (do_c) c0 = *reinterpret_cast<const uint32_t *>(&ss->c[li >> 3][4 + (li & 7)][12]);
                }
                if (__builtin_amdgcn_ballot_w64(any) != 0) {
                    if (do_l) {
                        int px[20];
                        unpack4(w0, px[0], px[1], px[2], px[3]);
                        unpack4(w1, px[4], px[5], px[6], px[7]);
                        unpack4(w2, px[8], px[9], px[10], px[11]);
                        unpack4(w3, px[12], px[13], px[14], px[15]);
                        unpack4(w4, px[16], px[17], px[18], px[19]);
                        const uint32_t bsp = Vbs[0], ab = Vab[0], tc = Vtc[0];
                        const int a1 = static_cast<int>((ab >> 16) & 255u), be1 = static_cast<int>(ab >> 24);
                        filter_edge<4, false>(px, static_cast<int>(bsp & 255u), static_cast<int>(ab & 255u), static_cast<int>((ab >> 8) & 255u), static_cast<int>(tc & 255u));
                        filter_edge<8, false>(px, static_cast<int>((bsp >> 8) & 255u), a1, be1, static_cast<int>((tc >> 8) & 255u));
                        filter_edge<12, false>(px, static_cast<int>((bsp >> 16) & 255u), a1, be1, static_cast<int>((tc >> 16) & 255u));
                        filter_edge<16, false>(px, static_cast<int>(bsp >> 24), a1, be1, static_cast<int>(tc >> 24));
                        w0 = pack4(px[0], px[1], px[2], px[3]), w1 = pack4(px[4], px[5], px[6], px[7]), w2 = pack4(px[8], px[9], px[10], px[11]);
                        w3 = pack4(px[12], px[13], px[14], px[15]), w4 = pack4(px[16], px[17], px[18], px[19]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (do_c) { // chroma: luma edges 0 and 2
                        int px[12];
                        unpack4(c0, px[0], px[1], px[2], px[3]);
                        unpack4(c1, px[4], px[5], px[6], px[7]);
                        unpack4(c2, px[8], px[9], px[10], px[11]);
                        const uint32_t bsp = Vbs[1], ab = Vab[1], tc = Vtc[1];
                        filter_edge<4, true>(px, static_cast<int>(bsp & 255u), static_cast<int>(ab & 255u), static_cast<int>((ab >> 8) & 255u), static_cast<int>(tc & 255u));
                        filter_edge<8, true>(px, static_cast<int>((bsp >> 16) & 255u), static_cast<int>((ab >> 16) & 255u), static_cast<int>(ab >> 24), static_cast<int>((tc >> 16) & 255u));
                        c0 = pack4(px[0], px[1], px[2], px[3]), c1 = pack4(px[4], px[5], px[6], px[7]), c2 = pack4(px[8], px[9], px[10], px[11]);
                    }
                }
                if (active) { // the line goes into the tile for the horizontal pass
                    if (do_l) {
                        *reinterpret_cast<uint32_t *>(&ss->y[4 + li][12]) = w0;
                        *reinterpret_cast<v4u *>(&ss->y[4 + li][16]) = v4u{w1, w2, w3, w4};
                    }
                    if (do_c) {
                        uint8_t *cr = &ss->c[li >> 3][4 + (li & 7)][4];
                        *reinterpret_cast<uint32_t *>(cr) = c0;
                        *reinterpret_cast<v2u *>(cr + 4) = v2u{c1, c2};
                    }
                }
            }
            WAVE_SYNC();
            STAMP(1);
            // ---- 3. hand-off of the rows above ----
            // 3a. columns 12..15 of the previous macroblock are final now: complete its bottom rows where they wait
            //     (the buffer for the sub-row below, or the ring slot for the group below), then publish the column
            const bool to_ring = sub == last_sub; // the group's last row feeds the next group, the others the sub-row below
            if (active && mbx > 0 && !last_row && li < 8 && (li < 4 ? do_l : do_c)) {
                const int cpl = (li >> 1) & 1, r = li & 1;
                GroupSlot *gl = &out_ring[(mbx - 1) % out_depth];
                uint32_t *dst;
                uint32_t v;
                if (li < 4)
                    dst = reinterpret_cast<uint32_t *>(to_ring ? &gl->y[li][12] : &ss->bot_y[li][12]), v = *reinterpret_cast<const uint32_t *>(&ss->y[16 + li][12]);
                else
                    dst = reinterpret_cast<uint32_t *>(to_ring ? &gl->c[cpl][r][4] : &ss->bot_c[cpl][r][4]), v = *reinterpret_cast<const uint32_t *>(&ss->c[cpl][10 + r][4]);
                *dst = v;
            }
            WAVE_SYNC();
            if (feeds_group) {
                const int xl = t - last_sub; // column of the group's last row in this step: columns 0 .. xl - 1 are final now
                if (to_global) { // the finished slot of column xl - 1 leaves as 24 granules
                    if (xl >= 1 && xl < wmb && gran)
                        __hip_atomic_store(xout + (xl - 1) * 24 + lane,
                                           (static_cast<unsigned long long>(epoch) << 32) | reinterpret_cast<const uint32_t *>(&out_ring[(xl - 1) % out_depth])[lane],
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else
                if (xl >= 1 && xl < wmb && lane == 0) __hip_atomic_store(&sh.prog[pc], xl, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // 3b. rows above this macroblock: from the sub-row above (same wavefront, written in the previous step and just
            //     completed), or -- sub-row 0 -- from the group above through its ring, once it says the column is final
            if (band_first) {
                if (t < wmb) {
                    // every granule of column t must carry this launch's epoch (the data is the flag); stragglers are re-read
                    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                    for (;;) {
                        const bool ok = !gran || static_cast<uint32_t>(pf >> 32) == epoch;
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                        __builtin_amdgcn_s_sleep(2);
                        if (gran) pf = __hip_atomic_load(xin + t * 24 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__builtin_amdgcn_s_memrealtime() - t_start > 400000000ull) { // 4 s at 100 MHz: report instead of hanging the GPU
                            if (lane == 0) atomicExch(xstatus, 0x5D000000u | static_cast<uint32_t>(g));
                            break;
                        }
                    }
                    if (gran) reinterpret_cast<uint32_t *>(in_stage)[lane] = static_cast<uint32_t>(pf);
                    if (t + 1 < wmb && gran) pf = __hip_atomic_load(xin + (t + 1) * 24 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // next step's slot
                }
            } else
            if (g > 0 && t < wmb)
                while (__hip_atomic_load(&sh.prog[pc_up], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < t + 1) __builtin_amdgcn_s_sleep(1);
            WAVE_SYNC();
            if (active && has_top && li < 8 && (li < 4 ? do_l : do_c)) {
                const int cpl = (li >> 1) & 1, r = li & 1;
                const GroupSlot *gs = &in_ring[(mbx > 0 ? mbx : 0) % in_depth];
                if (li < 4)
                    *reinterpret_cast<v4u *>(&ss->y[li][16]) = *reinterpret_cast<const v4u *>(sub > 0 ? sup->bot_y[li] : gs->y[li]);
                else
                    *reinterpret_cast<v2u *>(&ss->c[cpl][2 + r][8]) = *reinterpret_cast<const v2u *>(sub > 0 ? sup->bot_c[cpl][r] : gs->c[cpl][r]);
            }
            WAVE_SYNC();
            if (g > 0 && t < wmb && lane == 0) // the hand-off slot of column t has been copied: the group above may reuse it
                __hip_atomic_store(&sh.cons[pc], t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            STAMP(2);
            // ---- 4. horizontal edges: lane li = luma column li, then chroma (plane li >> 3, column li & 7) ----
            {
                const bool any = (Hbs[0] | Hbs[1]) != 0;
                if (__builtin_amdgcn_ballot_w64(any) != 0) {
                    if (any) {
                        if (do_l) {
                            int px[20];
#pragma unroll
                            for (int r = 0; r < 20; r++) px[r] = ss->y[r][16 + li];
                            const uint32_t bsp = Hbs[0], ab = Hab[0], tc = Htc[0];
                            const int a1 = static_cast<int>((ab >> 16) & 255u), be1 = static_cast<int>(ab >> 24);
                            filter_edge<4, false>(px, static_cast<int>(bsp & 255u), static_cast<int>(ab & 255u), static_cast<int>((ab >> 8) & 255u), static_cast<int>(tc & 255u));
                            filter_edge<8, false>(px, static_cast<int>((bsp >> 8) & 255u), a1, be1, static_cast<int>((tc >> 8) & 255u));
                            filter_edge<12, false>(px, static_cast<int>((bsp >> 16) & 255u), a1, be1, static_cast<int>((tc >> 16) & 255u));
                            filter_edge<16, false>(px, static_cast<int>(bsp >> 24), a1, be1, static_cast<int>(tc >> 24));
#pragma unroll
                            for (int r = 1; r < 19; r++) ss->y[r][16 + li] = static_cast<uint8_t>(px[r]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (do_c) {
                            const int cpl = li >> 3, i = li & 7;
                            int px[12];
                            px[0] = px[1] = 0;
#pragma unroll
                            for (int r = 2; r < 12; r++) px[r] = ss->c[cpl][r][8 + i];
                            const uint32_t bsp = Hbs[1], ab = Hab[1], tc = Htc[1];
                            filter_edge<4, true>(px, static_cast<int>(bsp & 255u), static_cast<int>(ab & 255u), static_cast<int>((ab >> 8) & 255u), static_cast<int>(tc & 255u));
                            filter_edge<8, true>(px, static_cast<int>((bsp >> 16) & 255u), static_cast<int>((ab >> 16) & 255u), static_cast<int>(ab >> 24), static_cast<int>((tc >> 16) & 255u));
#pragma unroll
                            for (int r = 3; r < 9; r++) ss->c[cpl][r][8 + i] = static_cast<uint8_t>(px[r]);
                        }
                    }
                }
            }
            WAVE_SYNC();
            STAMP(3);
            // ---- 5. results ----
            // Back-pressure first: the ring slot the group's last row is about to overwrite held column xl - depth of this
            // group; the group below must have consumed it.
            if (feeds_group && !to_global) { // (a slot that went to the global ring has been copied out: nothing to wait for)
                const int xl = t - last_sub;
                if (xl >= out_depth && xl < wmb)
                    while (__hip_atomic_load(&sh.cons[pc_dn], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < xl - out_depth + 1) __builtin_amdgcn_s_sleep(1);
            }
            if (active) {
                const bool row_end = mbx == wmb - 1;
                // finished bytes of this step.  Own rows: columns 12..15 of the macroblock to the left complete ITS register
                // slot, columns 0..11 of this one open a new slot.  Rows of the macroblock above (up lanes): all 16 columns.
                uint32_t l4 = 0, cl4 = 0;
                v4u own = z4;
                v2u cown = z2;
                if (do_l) {
                    l4 = *reinterpret_cast<const uint32_t *>(&ss->y[4 + li][12]);
                    own = *reinterpret_cast<const v4u *>(&ss->y[up_lane ? li - 12 : 4 + li][16]); // up lanes: tile rows 1..3 = rows -3..-1
                }
                if (do_c) {
                    const uint8_t *crp = &ss->c[li >> 3][upc_lane ? 3 : 4 + (li & 7)][4]; // chroma up lanes: row -1
                    cl4 = *reinterpret_cast<const uint32_t *>(crp);
                    cown = *reinterpret_cast<const v2u *>(crp + 4);
                }
                if (do_l && y_stores) {
                    const uint32_t yb = yout + mbx * 16;
                    if (up_lane)
                        GST16(py, yb, own); // a row of the macroblock above: all 16 columns are final
                    else {
                        if (mbx > 0)
                            GST16(py, yb - 4, (v4u{l4, own.x, own.y, own.z}));
                        else {
                            GST8(py, yb, (v2u{own.x, own.y}));
                            GST4(py, yb + 8, own.z);
                        }
                        if (row_end) GST4(py, yb + 12, own.w); // no macroblock to the right: the last columns are final too
                    }
                }
                if (do_c && c_stores) {
                    const uint32_t cb = cout + mbx * 8;
                    if (upc_lane)
                        GST8(py, cb, cown);
                    else {
                        if (mbx > 0)
                            GST8(py, cb - 4, (v2u{cl4, cown.x}));
                        else
                            GST4(py, cb, cown.x);
                        if (row_end) GST4(py, cb + 4, cown.y);
                    }
                }
                if (last_row && has_top) { // the up lanes own rows 13..15 here: rows -3..-1 of the macroblock above go out directly
                    if (li < 3 && do_l)
                        GST16(py, y_off + static_cast<uint32_t>(mby * 16 - 3 + li) * W + mbx * 16, *reinterpret_cast<const v4u *>(&ss->y[1 + li][16]));
                    else if (li >= 8 && li < 10 && do_c)
                        GST8(py, (li == 8 ? cb_off : cr_off) + static_cast<uint32_t>(mby * 8 - 1) * Wc + mbx * 8, *reinterpret_cast<const v2u *>(&ss->c[li - 8][3][8]));
                }
                // bottom rows of this macroblock (columns 12..15 still provisional unless the row ends here) for whoever is below
                if (!last_row && li < 8 && (li < 4 ? do_l : do_c)) {
                    const int cpl = (li >> 1) & 1, r = li & 1;
                    GroupSlot *gs = &out_ring[mbx % out_depth];
                    if (li < 4)
                        *reinterpret_cast<v4u *>(to_ring ? gs->y[li] : ss->bot_y[li]) = *reinterpret_cast<const v4u *>(&ss->y[16 + li][16]);
                    else
                        *reinterpret_cast<v2u *>(to_ring ? gs->c[cpl][r] : ss->bot_c[cpl][r]) = *reinterpret_cast<const v2u *>(&ss->c[cpl][10 + r][8]);
                }
            }
            WAVE_SYNC();
            if (to_global) {
                if (t - last_sub == wmb - 1 && gran)
                    __hip_atomic_store(xout + (wmb - 1) * 24 + lane,
                                       (static_cast<unsigned long long>(epoch) << 32) | reinterpret_cast<const uint32_t *>(&out_ring[(wmb - 1) % out_depth])[lane],
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else
            if (feeds_group && t - last_sub == wmb - 1 && lane == 0) // the last column of the group's last row is final without a right neighbour
                __hip_atomic_store(&sh.prog[pc], wmb, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            STAMP(4);
        }
#if defined(MI_DB_STATS)
        if (g == 0 && lane_v == 0)
            for (int k = 0; k < 12; k++) atomicAdd(xstatus + 8 + k, st_acc[k]);
#endif
    }
}


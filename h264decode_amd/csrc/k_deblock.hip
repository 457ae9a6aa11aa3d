// h264decode_amd/csrc/k_deblock.hip -- K5: in-loop deblocking filter (ITU-T H.264 8.7), gfx950; one workgroup per picture, two lines per lane.
//
// 8.7 is specified per macroblock in raster order (vertical edges left to right, then horizontal edges top to bottom), and the
// left-edge filter of MB(x+1,y) rewrites columns 13..15 of MB(x,y) AFTER MB(x,y)'s horizontal edges were filtered, so a
// whole-picture "all vertical, then all horizontal" pass is not bit-exact.  Every macroblock row is one serial chain (V0..V3 of MB x,
// its horizontal edges, V0 of MB x+1, ...), and the top edge of MB(x,y) needs rows 13..15 of MB(x,y-1) after V0 of MB(x+1,y-1).
// So a picture is a 2-D wavefront in which row y trails row y-1 by ONE macroblock, provided the vertical-edge pass of a step runs
// before the horizontal-edge pass of the same step.
//
// Mapping (round 5).  One workgroup owns a picture; a wavefront owns a GROUP of 8 consecutive macroblock rows ("sub-rows"), 8 lanes
// per macroblock -- 9 wavefronts and ONE round for 1080p.  At step t sub-row s works on macroblock column x = t - s.  The edge filters
// run on two lines at once in packed 16-bit arithmetic (k_deblock_pk.h): in the vertical-edge pass lane j owns luma rows 2j, 2j + 1 (one
// boundary-strength segment) and chroma row j of Cb and of Cr (one half each); in the horizontal-edge pass luma columns 2j, 2j + 1 and
// chroma column j.  The transposition between the passes goes through an LDS window of four macroblock columns per sub-row whose luma
// dwords are 2x2 sample blocks {Y(2r,2c), Y(2r+1,2c), Y(2r,2c+1), Y(2r+1,2c+1)}: the row-pair lane writes eight of them with two
// 16-byte stores (one byte permute each), the column-pair lane reads ten with one dword load each (an AND and a byte permute split a
// block into two packed column pairs).  Chroma dwords are {Cb(r,2k), Cb(r,2k+1), Cr(r,2k), Cr(r,2k+1)}.
// A step:
//   0. MB x-2 leaves for HBM straight from the window, 16 bytes per row: its rows -4..11 (rows 12..15 of the macroblock above, out of the
//      sub-row above's window, and its own rows 0..11); then the loads for the steps to come are issued;
//   1. vertical edges of MB x: 16 fresh columns from the prefetch registers (whole 64-byte lines, four macroblocks at a time, one step
//      ahead; the slot is a wave-uniform register index), columns 12..15 of MB x-1 from the window; results into the window;
//   2. MB x-1 is final now but for what the row below will do to its rows 13..15: the group's last sub-row copies its rows 12..15 into
//      the LDS ring of the group below and publishes the column (workgroup-scope release / acquire on two counters, as before), the
//      first sub-row takes column x of the group above;
//   3. horizontal edges of MB x on the window (rows -4..-1 = rows 12..15 of the sub-row above's window: no copy).
// Strengths and alpha / beta / tC0 come ready-made from k_dbprep (DbPrm), fetched one step ahead into registers; a lane derives its packed
// parameters with a handful of byte permutes (a permute IS the table lookup: bS selects its tC0 byte).
// No HBM access sits on the dependency path, and nothing is stored twice.
//
// Absent from the reference (only the slice-header fields are parsed: h264/slice.go:1021-1027).
#include <hip/hip_runtime.h>
#include "mi_kernels.h"
#include "k_deblock_pk.h"

#define WAVE_SYNC()                                            \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

struct Db8Shared { // followed in dynamic LDS by the wavefronts' windows and the hand-off rings
    int prog[MI_DEBLOCK8_MAX_GROUPS]; // per group: macroblock columns of its LAST row whose rows 12..15 are in the ring
    int cons[MI_DEBLOCK8_MAX_GROUPS]; // per group: hand-off slots consumed by its FIRST row
};
static_assert(sizeof(Db8Shared) <= MI_DEBLOCK8_HDR_BYTES, "LDS layout constants");

// Window of one sub-row: four macroblock columns ("slots", column x in slot x & 3).
//   luma   dword (slot, row pair rp, column pair i) at slot * 256 + rp * 32 + i * 4
//   chroma dword (slot, row r, column pair k)       at 1024 + slot * 128 + r * 16 + k * 4
// padded to 1568 bytes so that the windows of a wavefront's sub-rows start 8 banks apart (the column-pair lanes of a sub-row read 8
// consecutive dwords: eight sub-rows then cover the 32 banks twice).  A wavefront has nine windows: index 0 holds only rows 12..15 of
// the sub-row ABOVE its first one (taken from the ring), so that "the window above" is the same address arithmetic for every sub-row.
#define T_CHROMA 1024
#define T_BYTES MI_DEBLOCK8_TILE_BYTES
static_assert(MI_DEBLOCK8_WAVE_BYTES == 9 * T_BYTES + 2 * 8 * sizeof(DbPrm) && T_BYTES >= 1536 && (T_BYTES / 4) % 32 == 8, "LDS layout constants");
// ring slot: rows 12..15 of one macroblock column in window format: row pairs 6, 7 (32 bytes each), chroma rows 6, 7 (16 bytes each)
static_assert(MI_DEBLOCK_SLOT_BYTES == 96, "LDS layout constants");

typedef __attribute__((address_space(1))) uint8_t g8;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) v4u g_uint4;
typedef __attribute__((address_space(1))) v2u g_uint2;
#define GLD16(base, off) (*reinterpret_cast<const g_uint4 *>((base) + (off)))
#define GLD8(base, off) (*reinterpret_cast<const g_uint2 *>((base) + (off)))
#define GST16(base, off, v) (*reinterpret_cast<g_uint4 *>((base) + (off)) = (v))
#define GST8(base, off, v) (*reinterpret_cast<g_uint2 *>((base) + (off)) = (v))
// Loads are issued through inline assembly and waited for by ONE explicit s_waitcnt at the end of a step: gfx9 counts loads and stores in one
// in-order counter, and the compiler, which cannot see across the loop's back edge which registers a load may still be writing, guards their
// every use -- a guard behind freshly issued memory operations puts their whole round trip (18 k clocks measured) into the step.  The
// predicate is applied INSIDE the statement (EXEC narrowed and restored around the load), so that the statement sits in straight-line code
// with its destination as an in-out operand: the compiler then has no merge point at which to copy a register a load is still writing
// (with the load under an `if`, it did -- and handed the landing registers to other values).
#define ALD16M(dst, base, off, cond)                                                                                                               \
    do {                                                                                                                                           \
        unsigned long long sv_;                                                                                                                    \
        asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\ts_cbranch_execz 1f\n\tglobal_load_dwordx4 %[d], %[o], %[b]\n1:\n\ts_mov_b64 exec, %[sv]"       \
                     : [d] "+v"(dst), [sv] "=&s"(sv_)                                                                                              \
                     : [o] "v"(static_cast<uint32_t>(off)), [b] "s"(base), [m] "s"(__builtin_amdgcn_ballot_w64(cond))                              \
                     : "scc");                                                                                                                     \
    } while (0)
// The group loads of one phase (the step's t + 1 = K mod 8: sub-row K reloads -- its luma rows 0..7 and 8..15, the two halves of its chroma lines): four masked loads behind
// one scalar branch on the phase -- eight such statements stand in a step, one falls through.  Offsets and lane masks are the step's (the same operands in all
// eight); only the destinations differ.
#define ALD_PHASE(K, ga, gb, gc, gd, kq, oa, ob, oc, od, ml, mc, md, base)                                                                         \
    do {                                                                                                                                           \
        unsigned long long sv_;                                                                                                                    \
        asm volatile("s_cmp_lg_u32 %[q], " #K "\n\ts_cbranch_scc1 9f\n\ts_mov_b64 %[sv], exec\n\t"                                                 \
                     "s_and_b64 exec, %[sv], %[mL]\n\ts_cbranch_execz 1f\n\tglobal_load_dwordx4 %[dA], %[oA], %[b]\n\tglobal_load_dwordx4 %[dB], %[oB], %[b]\n1:\n\t" \
                     "s_and_b64 exec, %[sv], %[mC]\n\ts_cbranch_execz 2f\n\tglobal_load_dwordx4 %[dC], %[oC], %[b]\n2:\n\t"                         \
                     "s_and_b64 exec, %[sv], %[mD]\n\ts_cbranch_execz 3f\n\tglobal_load_dwordx4 %[dD], %[oD], %[b]\n3:\n\t"                         \
                     "s_mov_b64 exec, %[sv]\n9:"                                                                                                   \
                     : [dA] "+v"(ga), [dB] "+v"(gb), [dC] "+v"(gc), [dD] "+v"(gd), [sv] "=&s"(sv_)                                                 \
                     : [q] "s"(kq), [oA] "v"(oa), [oB] "v"(ob), [oC] "v"(oc), [oD] "v"(od), [mL] "s"(ml), [mC] "s"(mc), [mD] "s"(md), [b] "s"(base) \
                     : "scc");                                                                                                                     \
    } while (0)
typedef __attribute__((address_space(3))) uint8_t l8;
typedef __attribute__((address_space(3))) v4u l_uint4;
typedef __attribute__((address_space(3))) v2u l_uint2;
typedef __attribute__((address_space(3))) uint32_t l_uint1;
#define LLD16(off) (*reinterpret_cast<const l_uint4 *>(lds + (off)))
#define LLD8(off) (*reinterpret_cast<const l_uint2 *>(lds + (off)))
#define LLD4(off) (*reinterpret_cast<const l_uint1 *>(lds + (off)))
#define LST16(off, v) (*reinterpret_cast<l_uint4 *>(lds + (off)) = (v))
#define LST8(off, v) (*reinterpret_cast<l_uint2 *>(lds + (off)) = (v))
#define LST4(off, v) (*reinterpret_cast<l_uint1 *>(lds + (off)) = (v))
#define LST1(off, v) (lds[off] = static_cast<uint8_t>(v))
// keeps lane-dependent values from being hoisted out of the step loop
#define OPAQUE(x) asm volatile("" : "+v"(x))
#define PERM(hi, lo, sel) __builtin_amdgcn_perm(static_cast<uint32_t>(hi), static_cast<uint32_t>(lo), static_cast<uint32_t>(sel))
// diagnostic build (-DMI_DB_STATS): shader clocks per phase of the step loop, summed over one wavefront's steps, added to xstatus[8 + phase]
// by the wavefront of group 0 of every picture (tools/deblock_phase_probe.py)
#if defined(MI_DB_STATS)
#define STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += static_cast<uint32_t>(now_ - st_last); st_last = now_; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// byte k of w in both halves
__device__ __forceinline__ pk2 splat_byte(uint32_t w, int k) { return pk_from(PERM(0u, w, 0x0C000C00u + 0x00010001u * static_cast<uint32_t>(k))); }
// byte k of lo in the low half, byte k of hi in the high half
__device__ __forceinline__ pk2 pair_byte(uint32_t hi, uint32_t lo, int k) { return pk_from(PERM(hi, lo, 0x0C040C00u + 0x00010001u * static_cast<uint32_t>(k))); }
// all ones if byte k of w is not zero
__device__ __forceinline__ uint32_t byte_on(uint32_t w, int k) { return ((w >> (8 * k)) & 255u) ? ~0u : 0u; }

extern "C" __global__ void __launch_bounds__(MI_DEBLOCK8_MAX_WAVES * 64) k_deblock(const uint32_t *pic_list, const PicDesc *pics, const DbPrm *dbprm, int ring, int ring_last,
                                                                                   int last_bufs, uint32_t *xstatus) {
    extern __shared__ uint4 dyn_lds[];
    l8 *const lds = (l8 *)(reinterpret_cast<uint8_t *>(dyn_lds));
    const int nthreads = static_cast<int>(blockDim.x), nwaves = nthreads >> 6;
    Db8Shared &sh = *reinterpret_cast<Db8Shared *>(dyn_lds);
    const int tid = static_cast<int>(threadIdx.x), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane_v = tid & 63;
    const PicDesc *pd = &pics[pic_list[blockIdx.x]];
    const int wmb = static_cast<int>(pd->wmb), hmb = static_cast<int>(pd->hmb);
    // the picture's place in its frame slot (PicDesc): W = bytes from one luma row of the PICTURE to the next (a field picture lives in the
    // rows of its parity: twice the frame's pitch, first row y_off bytes in); offsets are relative to the slot's first byte
    const int W = static_cast<int>(pd->pitch), Wc = W / 2;
    g8 *const py = (g8 *)(pd->pool_base + static_cast<uint64_t>(pd->slot) * pd->slot_bytes);
    const uint32_t y_off = pd->field == 2 ? pd->pitch >> 1 : 0u;
    const uint32_t cb_off = pd->plane + (y_off >> 1), cr_off = cb_off + (pd->plane >> 2);
    for (int i = tid; i < MI_DEBLOCK8_MAX_GROUPS; i += nthreads) sh.prog[i] = 0, sh.cons[i] = 0;
    __syncthreads();
    const g8 *const prms = (const g8 *)(dbprm + pd->mb_base);
    const int ngroups = (hmb + 7) >> 3;
    const uint32_t rings_off = MI_DEBLOCK8_HDR_BYTES + static_cast<uint32_t>(nwaves) * MI_DEBLOCK8_WAVE_BYTES; // region r (written by the groups of wavefront r) starts at r * ring slots
    const v4u z4 = v4u{0u, 0u, 0u, 0u};
    for (int g = wave; g < ngroups; g += nwaves) {
        int lane = lane_v;
        OPAQUE(lane);
        const int last_sub = min(7, hmb - 1 - g * 8); // last valid sub-row of this group
        const bool feeds_group = g + 1 < ngroups;     // this group's last row hands its bottom rows to group g + 1
        // hand-off rings: the one this group writes (region `wave`) and the one it reads (written by group g - 1)
        // (the last wavefront's region holds whole rows and, from three rounds on, one buffer per round parity: see mi_deblock8_plan)
        const bool out_last = wave == nwaves - 1;
        const int out_depth = out_last ? ring_last : ring;
        const uint32_t out_ring = rings_off + static_cast<uint32_t>(wave * ring + (out_last ? ((g / nwaves) % last_bufs) * ring_last : 0)) * MI_DEBLOCK_SLOT_BYTES;
        const int in_wave = (g + nwaves - 1) % nwaves;
        const bool in_last = in_wave == nwaves - 1;
        const int in_depth = in_last ? ring_last : ring;
        const uint32_t in_ring = rings_off + static_cast<uint32_t>(in_wave * ring + (in_last && g > 0 ? (((g - 1) / nwaves) % last_bufs) * ring_last : 0)) * MI_DEBLOCK_SLOT_BYTES;
        // ---- memory traffic is COOPERATIVE: a vector memory instruction costs what its lanes touch in distinct cache lines (the CU's L1 looks up one line
        // per clock: with a lane per row, 64 lines per instruction, 9 wavefronts spent 11 k of a step's 18 k clocks issuing them), so whole
        // wavefronts move blocks of one sub-row between HBM and its LDS window, eight / four (loads) or two (stores) adjacent lanes per row:
        //   luma load    GA[k] / GB[k], rows 0..7 / 8..15 of sub-row k, EIGHT macroblock columns at once = whole 128-byte lines: lane = (row L >> 3, column L & 7),
        //                16 bytes; landed pieces enter the window one column per step (the slot of column x + 1 is free from the end of step x on), raw rows --
        //                rows 2j, 2j + 1 are the 32 bytes of row pair j, so the vertical pass converts its own 32 bytes in place;
        //   chroma load  GC[k] / GD[k], both planes' eight rows of sub-row k, SIXTEEN columns at once = whole lines, as two instructions of half lines in the same step:
        //                lane = (plane L >> 5, row (L >> 2) & 7, column pair L & 3), 16 bytes = two macroblocks' eight samples; issued with every other luma group;
        //   stores       column pairs {x - 3, x - 2} of the sub-rows whose x is odd: luma lane = (sub-row, row (L & 31) >> 1, column L & 1),
        //                chroma lane = (sub-row, plane, row), 16 bytes = both columns; a sub-row stores ITS OWN rows 0..15 (what the row below did to
        //                rows 13..15 happened in this window), the first sub-row also rows 12..15 of the group above out of window 0, the last
        //                sub-row of a feeding group not its rows 12..15 (they went down the ring);
        //   DbPrm        40 lanes x 16 bytes = the records of the eight macroblocks of the next step, through a double-buffered LDS stage.
        const uint32_t tile0 = MI_DEBLOCK8_HDR_BYTES + static_cast<uint32_t>(wave) * MI_DEBLOCK8_WAVE_BYTES; // window k + 1 = sub-row k
        const uint32_t stage0 = tile0 + 9 * T_BYTES;                                                           // 2 x 8 x sizeof(DbPrm)
        const int rows_here = min(8, hmb - g * 8);                                                              // valid sub-rows of this group
        // luma load lane
        const uint32_t ll_src = y_off + static_cast<uint32_t>(g * 128 + (lane >> 3)) * W + (lane & 7) * 16; // + (k * 16 [+ 8]) W + column base * 16
        // chroma load lane
        const uint32_t lc_src = ((lane & 32) ? cr_off : cb_off) + static_cast<uint32_t>(g * 64 + ((lane >> 2) & 7)) * Wc + (lane & 3) * 16; // + k * 8 Wc + column base * 8 [+ 64: the line's other half]
        // luma store lane (instruction i: sub-rows par + 4 i and par + 4 i + 2)
        // chroma store lane (sub-rows par, par + 2, par + 4, par + 6)
        // DbPrm lane
        const int lp_sub = lane < 40 ? lane / 5 : 7, lp_piece = lane < 40 ? lane % 5 : 0;
        const uint32_t lp_row = static_cast<uint32_t>(min(g * 8 + lp_sub, hmb - 1)) * static_cast<uint32_t>(wmb);
        v4u GA0 = z4, GA1 = z4, GA2 = z4, GA3 = z4, GA4 = z4, GA5 = z4, GA6 = z4, GA7 = z4; // sub-row k: rows 0..7 of eight macroblock columns (whole 128-byte lines)
        v4u GB0 = z4, GB1 = z4, GB2 = z4, GB3 = z4, GB4 = z4, GB5 = z4, GB6 = z4, GB7 = z4; // ... rows 8..15
        v4u GC0 = z4, GC1 = z4, GC2 = z4, GC3 = z4, GC4 = z4, GC5 = z4, GC6 = z4, GC7 = z4; // sub-row k: both chroma planes' eight rows, columns 0..7 of SIXTEEN (the first half of the 128-byte lines)
        v4u GD0 = z4, GD1 = z4, GD2 = z4, GD3 = z4, GD4 = z4, GD5 = z4, GD6 = z4, GD7 = z4; // ... columns 8..15 (the other half, asked for in the same step)
        v4u GP = z4;
        // every load issued so far has landed (the one wait on vector memory of a step, at its end: what it waits for was issued at the step's top)
        auto loads_landed = [&]() {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(GA0), "+v"(GA1), "+v"(GA2), "+v"(GA3), "+v"(GA4), "+v"(GA5), "+v"(GA6), "+v"(GA7), "+v"(GB0), "+v"(GB1), "+v"(GB2), "+v"(GB3), "+v"(GB4),
                         "+v"(GB5), "+v"(GB6), "+v"(GB7));
            asm volatile("" : "+v"(GC0), "+v"(GC1), "+v"(GC2), "+v"(GC3), "+v"(GC4), "+v"(GC5), "+v"(GC6), "+v"(GC7), "+v"(GD0), "+v"(GD1), "+v"(GD2), "+v"(GD3), "+v"(GD4), "+v"(GD5),
                         "+v"(GD6), "+v"(GD7), "+v"(GP));
        };
        // the ring this group writes was last used by the group `reuse` groups earlier: that group's reader must be through with it
        // every wait on another wavefront gives up after 4 s of s_memrealtime and says so through the status word (H264MI_EDECODE) instead of hanging the GPU
        auto wait_for = [&](int *ctr, int want) {
            if (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) return;
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t_start > 400000000ull) { // 100 MHz
                    if (lane_v == 0) atomicExch(xstatus, 0x5D800000u | static_cast<uint32_t>(g));
                    break;
                }
            }
        };
        const int reuse = out_last ? nwaves * last_bufs : nwaves;
        if (g >= reuse && feeds_group) wait_for(&sh.cons[g - reuse + 1], wmb);
        const int t_last = wmb + 9; // sub-row s: loads from step s - 1 on, columns in steps s .. s + wmb - 1, the last column pair's output up to three steps later
#if defined(MI_DB_STATS)
        uint32_t st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const uint32_t st_t0 = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
        unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
        for (int t = -1; t <= t_last; t++) {
#if defined(MI_DB_PRIO)
            // Measured, not on (profiles/r05_k5_group_timeline.txt).  Two wavefronts share a SIMD, and its arbiter serves the OLDER one first: the wavefronts of groups
            // 0..3 take 4.6 us a step, those of groups 4..7 7 us -- and a picture is a chain of groups.  Taking turns at the higher issue priority (a step up, a step
            // down, the SIMD's two wavefronts in opposite phase) does share the SIMD evenly (5.2 / 6.0 us; group 7 ends at 1048 instead of 1140 us), but wavefront 0 then
            // comes to the ninth group of a 1080p picture 50 us later, and that group's 131 steps, not the chain, end the launch (1.17 against 1.13 ms).
            if ((t + (wave >> 2)) & 1)
                __builtin_amdgcn_s_setprio(2);
            else
                __builtin_amdgcn_s_setprio(0);
#endif
            STAMP(5); // loop control + whatever the compiler moved across the step boundary
#if defined(MI_DB_STATS)
            if (blockIdx.x == 0 && lane_v == 0 && g < 12 && t >= 0 && (t & 15) == 0 && (t >> 4) < 8) xstatus[64 + 8 * g + (t >> 4)] = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
#endif
            int lane = lane_v;
            OPAQUE(lane);
            const int s = lane >> 3, j = lane & 7;
            const int mby = g * 8 + s;
            const bool row_ok = mby < hmb, has_top = mby > 0;
            const int mbx = t - s;
            // (the lanes' roles in the cooperative transfers are re-derived from the lane number in every step: as loop invariants their predicates would sit in scalar
            // register pairs for the whole kernel -- which the compiler then spills into VGPR lanes)
            const int ll_row = lane >> 3, ll_col = lane & 7;
            const uint32_t ll_dst = tile0 + T_BYTES + (ll_col & 3) * 256 + ll_row * 16;                // + k * T_BYTES [+ 128: rows 8..15]  (slot = column & 3)
            const int lc_rr = lane >> 2, lc_q = lane & 3; // (plane, row) 0..15, column pair 0..3 of a half line's eight columns
            const uint32_t lc_dst = tile0 + T_BYTES + T_CHROMA + (lc_rr & 7) * 16 + (lc_rr >> 3) * 8;                                             // + k * T_BYTES + slot * 128
            const int sl_half = lane >> 5, sl_row = (lane & 31) >> 1, sl_col = lane & 1;
            const uint32_t sl_sel = (sl_row & 1) ? 0x07050301u : 0x06040200u; // this row of a row pair's 2x2 blocks
            const int sc_q = lane >> 4, sc_plane = (lane >> 3) & 1, sc_row = lane & 7;
            const uint32_t sc_sel = sc_plane ? 0x07060302u : 0x05040100u;
            const bool active = row_ok && mbx >= 0 && mbx < wmb;
            // LDS addresses: this sub-row's window (index s + 1 of the wavefront's nine), the window above, the slots of columns x and x - 1
            const uint32_t tile = MI_DEBLOCK8_HDR_BYTES + static_cast<uint32_t>(wave) * MI_DEBLOCK8_WAVE_BYTES + static_cast<uint32_t>(s + 1) * T_BYTES;
            const uint32_t sx = static_cast<uint32_t>(mbx) & 3u, spv = static_cast<uint32_t>(mbx - 1) & 3u;
            const uint32_t own_l = tile + sx * 256, prev_l = tile + spv * 256, own_c = tile + T_CHROMA + sx * 128, prev_c = tile + T_CHROMA + spv * 128;
            // this step's parameters: the macroblock's DbPrm out of the stage the previous step filled -- this lane's strengths (one dword per direction) and the three planes' blocks
            const uint32_t stage = stage0 + static_cast<uint32_t>(t & 1) * (8 * static_cast<uint32_t>(sizeof(DbPrm))) + s * static_cast<uint32_t>(sizeof(DbPrm));
            const v2u bs = LLD8(stage + (j >> 1) * 8); // (the planes' blocks are read where a pass needs them: 12 registers less across the step)
            STAMP(0);
            // ---- 0. the step's vector memory operations in one burst (the one wait for them is at the end of the step) ----
            // 0a. loads: the ONE sub-row whose column x + 1 starts a group of eight -- whole 128-byte lines, so that a line is fetched once (with groups of four the halves of
            // a luma line, the quarters of a chroma line, were asked for four steps apart, and the 32 pictures of an XCD keep more lines in flight than its 4 MB of L2 hold:
            // 2.2 x the bytes).  Luma: eight columns = a line; chroma: SIXTEEN columns = a line, every other time.
            {
                const int k8 = (t + 1) & 7, cb8 = t - k8 + 1;
                const bool ok8 = k8 < rows_here && cb8 >= 0 && cb8 < wmb;
                const unsigned long long m_l = ok8 ? __builtin_amdgcn_ballot_w64(ll_col < wmb - cb8) : 0ull;
                const bool okc = ok8 && !(cb8 & 8);
                const unsigned long long m_c = okc ? __builtin_amdgcn_ballot_w64(2 * lc_q < wmb - cb8) : 0ull, m_d = okc ? __builtin_amdgcn_ballot_w64(2 * lc_q + 8 < wmb - cb8) : 0ull;
                const uint32_t o_l = ll_src + __umul24(k8 * 16, W) + cb8 * 16, o_h = o_l + 8 * static_cast<uint32_t>(W);
                const uint32_t o_c = lc_src + __umul24(k8 * 8, Wc) + cb8 * 8, o_d = o_c + 64;
                ALD_PHASE(0, GA0, GB0, GC0, GD0, k8, o_l, o_h, o_c, o_d, m_l, m_c, m_d, py);
                ALD_PHASE(1, GA1, GB1, GC1, GD1, k8, o_l, o_h, o_c, o_d, m_l, m_c, m_d, py);
                ALD_PHASE(2, GA2, GB2, GC2, GD2, k8, o_l, o_h, o_c, o_d, m_l, m_c, m_d, py);
                ALD_PHASE(3, GA3, GB3, GC3, GD3, k8, o_l, o_h, o_c, o_d, m_l, m_c, m_d, py);
                ALD_PHASE(4, GA4, GB4, GC4, GD4, k8, o_l, o_h, o_c, o_d, m_l, m_c, m_d, py);
                ALD_PHASE(5, GA5, GB5, GC5, GD5, k8, o_l, o_h, o_c, o_d, m_l, m_c, m_d, py);
                ALD_PHASE(6, GA6, GB6, GC6, GD6, k8, o_l, o_h, o_c, o_d, m_l, m_c, m_d, py);
                ALD_PHASE(7, GA7, GB7, GC7, GD7, k8, o_l, o_h, o_c, o_d, m_l, m_c, m_d, py);
                // DbPrm of the next step's macroblocks (a clamped address where there is none: the stage entry is never used then)
                ALD16M(GP, prms, __umul24(lp_row + static_cast<uint32_t>(min(max(t - lp_sub + 1, 0), wmb - 1)), static_cast<uint32_t>(sizeof(DbPrm))) + lp_piece * 16, lane < 40); // (a picture has at most 2^18 macroblocks)
            }
            STAMP(7);
            // 0b. stores: the column pairs {x - 3, x - 2} of the sub-rows whose x is odd (final since the vertical pass of step x - 1)
            {
                const int par = (t + 1) & 1;
                // (the window reads of all three instructions first, unpredicated -- their addresses are inside the wavefront's windows whatever x is --,
                // so that one LDS round trip serves them all; only the stores are predicated)
                int kl[2], cl[2];
                v4u wl0[2], wl1[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    kl[i] = par + 4 * i + 2 * sl_half, cl[i] = t - kl[i] - 3 + sl_col;
                    const uint32_t src = tile0 + __umul24(kl[i] + 1, T_BYTES) + (static_cast<uint32_t>(cl[i]) & 3u) * 256 + (sl_row >> 1) * 32;
                    wl0[i] = LLD16(src), wl1[i] = LLD16(src + 16);
                }
                const int kc = par + 2 * sc_q, cc = t - kc - 3;
                const uint32_t srcc = tile0 + __umul24(kc + 1, T_BYTES) + T_CHROMA + sc_row * 16;
                const v4u wa = LLD16(srcc + (static_cast<uint32_t>(cc) & 3u) * 128), wb = LLD16(srcc + (static_cast<uint32_t>(cc + 1) & 3u) * 128);
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int k = kl[i], col = cl[i];
                    if (k < rows_here && col - sl_col >= 0 && col < wmb && !(sl_row >= 12 && feeds_group && k == last_sub))
                        GST16(py, y_off + __umul24((g * 8 + k) * 16 + sl_row, W) + col * 16,
                              (v4u{PERM(wl0[i].y, wl0[i].x, sl_sel), PERM(wl0[i].w, wl0[i].z, sl_sel), PERM(wl1[i].y, wl1[i].x, sl_sel), PERM(wl1[i].w, wl1[i].z, sl_sel)}));
                }
                if (kc < rows_here && cc >= 0 && cc < wmb && !(sc_row == 7 && feeds_group && kc == last_sub)) {
                    const uint32_t o = (sc_plane ? cr_off : cb_off) + __umul24((g * 8 + kc) * 8 + sc_row, Wc) + cc * 8;
                    if (cc + 1 < wmb)
                        GST16(py, o, (v4u{PERM(wa.y, wa.x, sc_sel), PERM(wa.w, wa.z, sc_sel), PERM(wb.y, wb.x, sc_sel), PERM(wb.w, wb.z, sc_sel)}));
                    else
                        GST8(py, o, (v2u{PERM(wa.y, wa.x, sc_sel), PERM(wa.w, wa.z, sc_sel)}));
                }
                // rows 12..15 (chroma row 7) of the group above, completed in window 0 by this group's first sub-row: lanes 0..7 luma, 8..9 chroma
                if (g > 0 && par == 0 && t >= 3 && lane < 10) { // (x of the first sub-row = t, odd)
                    const int col = t - 3 + (lane < 8 ? (lane & 1) : 0);
                    if (col < wmb) {
                        if (lane < 8) {
                            const int r = lane >> 1; // rows -4 + r
                            const uint32_t src = tile0 + (static_cast<uint32_t>(col) & 3u) * 256 + 192 + (r >> 1) * 32, sel = (r & 1) ? 0x07050301u : 0x06040200u;
                            const v4u w0 = LLD16(src), w1 = LLD16(src + 16);
                            GST16(py, y_off + __umul24(g * 128 - 4 + r, W) + col * 16, (v4u{PERM(w0.y, w0.x, sel), PERM(w0.w, w0.z, sel), PERM(w1.y, w1.x, sel), PERM(w1.w, w1.z, sel)}));
                        } else {
                            const uint32_t src = tile0 + T_CHROMA + 112, sel = lane == 9 ? 0x07060302u : 0x05040100u;
                            const v4u wa = LLD16(src + (static_cast<uint32_t>(col) & 3u) * 128), wb = LLD16(src + (static_cast<uint32_t>(col + 1) & 3u) * 128);
                            const uint32_t o = (lane == 9 ? cr_off : cb_off) + __umul24(g * 64 - 1, Wc) + col * 8;
                            if (col + 1 < wmb)
                                GST16(py, o, (v4u{PERM(wa.y, wa.x, sel), PERM(wa.w, wa.z, sel), PERM(wb.y, wb.x, sel), PERM(wb.w, wb.z, sel)}));
                            else
                                GST8(py, o, (v2u{PERM(wa.y, wa.x, sel), PERM(wa.w, wa.z, sel)}));
                        }
                    }
                }
            }
            STAMP(4);
            // ---- hand-off, first half: the group's first row takes rows 12..15 of column t of the group above out of its ring, once that says the column is final (the group
            // above is eight steps ahead: no wait in the steady state); the copy needs no synchronisation of its own -- the one behind the vertical pass stands between it and its readers
            if (g > 0 && t >= 0 && t < wmb) {
                wait_for(&sh.prog[g - 1], t + 1);
                if (s == 0 && j < 6) {
                    const uint32_t above = tile - T_BYTES;
                    const uint32_t dst = j < 4 ? above + sx * 256 + 192 + j * 16 : above + T_CHROMA + sx * 128 + 96 + (j - 4) * 16;
                    LST16(dst, LLD16(in_ring + static_cast<uint32_t>(t % in_depth) * MI_DEBLOCK_SLOT_BYTES + j * 16));
                }
            }
            // ---- 1. vertical edges: lane j = luma rows 2j, 2j + 1, then chroma row j of Cb | Cr ----
            if (active) {
                // a plane's block: aL bL aI bI | aT bT tL1 tL2 | tL3 tI1 tI2 tI3 | tT1 tT2 tT3 pad  (a / b: alpha / beta of the left-edge, inner, top-edge QP average; tKb: tC0 for bS b).
                // bS -> position of its tC0 byte in {y, z}: one permute maps the four strengths, a second one fetches the four bytes.
                const uint32_t bsv = bs.x;
                const uint32_t tsel = PERM(0x0403020Cu, 0x0706050Cu, bsv + 4u); // edge 0: the left-edge row (bytes 2..4 of {y, z}), inner edges: bytes 5..7; bS 0: a zero
                const bool left = mbx > 0;
                {
                    pk2 v[20]; // luma columns -4..15 of rows 2j | 2j + 1
                    {
                        const v4u in_a = LLD16(own_l + j * 32), in_b = LLD16(own_l + j * 32 + 16); // the raw rows 2j, 2j + 1 (the loads' landing place: the bytes of row pair j)
#pragma unroll
                        for (int k = 0; k < 4; k++)
#pragma unroll
                            for (int m = 0; m < 4; m++) v[4 + 4 * k + m] = pair_byte(in_b[k], in_a[k], m);
                    }
                    uint32_t w6 = 0, w7 = 0;
                    bool f0 = false;
                    if (__builtin_amdgcn_ballot_w64(bsv != 0) != 0) {
                        const v4u p0 = LLD16(stage + 32);
                        const uint32_t tc4 = PERM(p0.z, p0.y, tsel);
                        const pk2 aL = splat_byte(p0.x, 0), bL = splat_byte(p0.x, 1), aI = splat_byte(p0.x, 2), bI = splat_byte(p0.x, 3);
                        if (left) {
                            const v2u lw = LLD8(prev_l + j * 32 + 24); // columns 12..15 of the previous macroblock, after its horizontal pass
                            w6 = lw.x, w7 = lw.y;
                        }
                        v[0] = pk_from(PERM(0u, w6, 0x0C010C00u)), v[1] = pk_from(PERM(0u, w6, 0x0C030C02u));
                        v[2] = pk_from(PERM(0u, w7, 0x0C010C00u)), v[3] = pk_from(PERM(0u, w7, 0x0C030C02u));
                        f0 = pk_luma_edge<true>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], aL, bL, splat_byte(tc4, 0), byte_on(bsv, 0), (bsv & 255u) == 4u ? ~0u : 0u);
                        pk_luma_edge<false>(v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], aI, bI, splat_byte(tc4, 1), byte_on(bsv, 1), 0u);
                        pk_luma_edge<false>(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], aI, bI, splat_byte(tc4, 2), byte_on(bsv, 2), 0u);
                        pk_luma_edge<false>(v[12], v[13], v[14], v[15], v[16], v[17], v[18], v[19], aI, bI, splat_byte(tc4, 3), byte_on(bsv, 3), 0u);
                        if (f0 && left) LST8(prev_l + j * 32 + 24, (v2u{PERM(pk_bits(v[1]), pk_bits(v[0]), 0x06040200u), PERM(pk_bits(v[3]), pk_bits(v[2]), 0x06040200u)}));
                    }
                    v4u o0, o1; // the lines go into the window as 2x2 blocks of the two rows
#pragma unroll
                    for (int i = 0; i < 4; i++) o0[i] = PERM(pk_bits(v[5 + 2 * i]), pk_bits(v[4 + 2 * i]), 0x06040200u), o1[i] = PERM(pk_bits(v[13 + 2 * i]), pk_bits(v[12 + 2 * i]), 0x06040200u);
                    LST16(own_l + j * 32, o0), LST16(own_l + j * 32 + 16, o1);
                }
                __builtin_amdgcn_sched_barrier(0);
                // chroma: luma edges 0 and 2; the low half is Cb, the high half Cr
                const uint32_t bsc = bsv & 0x00FF00FFu;
                pk2 cv[10]; // chroma columns -2..7 of Cb | Cr
                {
                    const v4u in_c = LLD16(own_c + j * 16); // row j as it landed: Cb | Cr
#pragma unroll
                    for (int k = 0; k < 2; k++)
#pragma unroll
                        for (int m = 0; m < 4; m++) cv[2 + 4 * k + m] = pair_byte(in_c[2 + k], in_c[k], m);
                }
                if (__builtin_amdgcn_ballot_w64(bsc != 0) != 0) {
                    const v4u p1 = LLD16(stage + 48), p2 = LLD16(stage + 64);
                    const uint32_t tcb4 = PERM(p1.z, p1.y, tsel), tcr4 = PERM(p2.z, p2.y, tsel);
                    uint32_t w3 = 0;
                    if (left) w3 = LLD4(prev_c + j * 16 + 12);
                    cv[0] = pk_from(w3 & 0x00FF00FFu), cv[1] = pk_from(PERM(0u, w3, 0x0C030C01u));
                    const bool f0 = pk_chroma_edge<true>(cv[0], cv[1], cv[2], cv[3], pair_byte(p2.x, p1.x, 0), pair_byte(p2.x, p1.x, 1), pair_byte(tcr4, tcb4, 0) + pk_splat(1), byte_on(bsv, 0),
                                                         (bsv & 255u) == 4u ? ~0u : 0u);
                    pk_chroma_edge<false>(cv[4], cv[5], cv[6], cv[7], pair_byte(p2.x, p1.x, 2), pair_byte(p2.x, p1.x, 3), pair_byte(tcr4, tcb4, 2) + pk_splat(1), byte_on(bsv, 2), 0u);
                    if (f0 && left) LST4(prev_c + j * 16 + 12, pk_bits(cv[0]) | (pk_bits(cv[1]) << 8));
                }
                v4u o;
#pragma unroll
                for (int k = 0; k < 4; k++) o[k] = PERM(pk_bits(cv[3 + 2 * k]), pk_bits(cv[2 + 2 * k]), 0x06020400u);
                LST16(own_c + j * 16, o);
            }
            WAVE_SYNC();
            STAMP(1);
            // ---- 2. hand-off, second half: the group's last row -- rows 12..15 of column xl - 1 are final but for the row below: into the ring of the group below
            //     (six 16-byte pieces in window format), then the column is published (the release store waits for the wavefront's LDS writes).  Back-pressure
            //     first: the slot held column xl - 1 - depth, which the group below must have consumed.
            if (g > 0 && t >= 0 && t < wmb && lane == 0) // (the slot of column t was copied at the top of the step: the group above may reuse it)
                __hip_atomic_store(&sh.cons[g], t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (feeds_group) {
                const int xl = t - last_sub; // column of the group's last row in this step
                if (xl >= 1 && xl <= wmb) {
                    const int c = xl - 1;
                    if (c >= out_depth) wait_for(&sh.cons[g + 1], c - out_depth + 1);
                    if (s == last_sub && j < 6) {
                        const uint32_t src = j < 4 ? tile + spv * 256 + 192 + j * 16 : tile + T_CHROMA + spv * 128 + 96 + (j - 4) * 16;
                        LST16(out_ring + static_cast<uint32_t>(c % out_depth) * MI_DEBLOCK_SLOT_BYTES + j * 16, LLD16(src));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the copy above is other lanes' work: keep it in front of lane 0's store)
                    if (lane == 0) __hip_atomic_store(&sh.prog[g], xl, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            STAMP(2);
            // ---- 3. horizontal edges: lane j = luma columns 2j, 2j + 1, then chroma column j of Cb | Cr ----
            if (active) {
                const uint32_t bsh = bs.y;
                const uint32_t above = tile - T_BYTES;
                const uint32_t tsel = PERM(0x0605040Cu, 0x0302010Cu, bsh + 4u); // edge 0: the top-edge row (bytes 0..2 of w), inner edges: bytes 1..3 of z
                if (__builtin_amdgcn_ballot_w64(bsh != 0) != 0) {
                    const v4u p0 = LLD16(stage + 32);
                    const uint32_t tc4 = PERM(p0.w, p0.z, tsel);
                    const pk2 aT = splat_byte(p0.y, 0), bT = splat_byte(p0.y, 1), aI = splat_byte(p0.x, 2), bI = splat_byte(p0.x, 3);
                    pk2 h[20]; // rows -4..15 of columns 2j | 2j + 1
                    uint32_t wa6 = 0, wa7 = 0;
                    if (has_top) wa6 = LLD4(above + sx * 256 + 192 + j * 4), wa7 = LLD4(above + sx * 256 + 224 + j * 4);
                    h[0] = pk_from(wa6 & 0x00FF00FFu), h[1] = pk_from(PERM(0u, wa6, 0x0C030C01u));
                    h[2] = pk_from(wa7 & 0x00FF00FFu), h[3] = pk_from(PERM(0u, wa7, 0x0C030C01u));
#pragma unroll
                    for (int rp = 0; rp < 8; rp++) {
                        const uint32_t w = LLD4(own_l + rp * 32 + j * 4);
                        h[4 + 2 * rp] = pk_from(w & 0x00FF00FFu), h[5 + 2 * rp] = pk_from(PERM(0u, w, 0x0C030C01u));
                    }
                    const bool f0 = pk_luma_edge<true>(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], aT, bT, splat_byte(tc4, 0), byte_on(bsh, 0), (bsh & 255u) == 4u ? ~0u : 0u);
                    const bool f1 = pk_luma_edge<false>(h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], aI, bI, splat_byte(tc4, 1), byte_on(bsh, 1), 0u);
                    const bool f2 = pk_luma_edge<false>(h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15], aI, bI, splat_byte(tc4, 2), byte_on(bsh, 2), 0u);
                    const bool f3 = pk_luma_edge<false>(h[12], h[13], h[14], h[15], h[16], h[17], h[18], h[19], aI, bI, splat_byte(tc4, 3), byte_on(bsh, 3), 0u);
                    // edge e changed rows 4e - 3 .. 4e + 2: row pairs 2e - 2 .. 2e + 1 (pair -2, -1 = pairs 6, 7 of the window above)
                    if (f0 && has_top) {
                        LST4(above + sx * 256 + 192 + j * 4, pk_bits(h[0]) | (pk_bits(h[1]) << 8));
                        LST4(above + sx * 256 + 224 + j * 4, pk_bits(h[2]) | (pk_bits(h[3]) << 8));
                    }
                    const bool wr[8] = {f0 || f1, f0 || f1, f1 || f2, f1 || f2, f2 || f3, f2 || f3, f3, f3};
#pragma unroll
                    for (int rp = 0; rp < 8; rp++)
                        if (wr[rp]) LST4(own_l + rp * 32 + j * 4, pk_bits(h[4 + 2 * rp]) | (pk_bits(h[5 + 2 * rp]) << 8));
                }
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t bsc = bsh & 0x00FF00FFu;
                if (__builtin_amdgcn_ballot_w64(bsc != 0) != 0) {
                    const v4u p1 = LLD16(stage + 48), p2 = LLD16(stage + 64);
                    const uint32_t tcb4 = PERM(p1.w, p1.z, tsel), tcr4 = PERM(p2.w, p2.z, tsel);
                    const uint32_t csel = (j & 1) ? 0x0C030C01u : 0x0C020C00u; // this lane's column of a dword's column pair: Cb | Cr
                    const uint32_t ca = above + T_CHROMA + sx * 128 + (j >> 1) * 4, co = own_c + (j >> 1) * 4;
                    pk2 c[10]; // rows -2..7
                    uint32_t wa = 0, wb = 0;
                    if (has_top) wa = LLD4(ca + 96), wb = LLD4(ca + 112);
                    c[0] = pk_from(PERM(0u, wa, csel)), c[1] = pk_from(PERM(0u, wb, csel));
#pragma unroll
                    for (int r = 0; r < 8; r++) c[2 + r] = pk_from(PERM(0u, LLD4(co + r * 16), csel));
                    const bool f0 = pk_chroma_edge<true>(c[0], c[1], c[2], c[3], pair_byte(p2.y, p1.y, 0), pair_byte(p2.y, p1.y, 1), pair_byte(tcr4, tcb4, 0) + pk_splat(1), byte_on(bsh, 0),
                                                         (bsh & 255u) == 4u ? ~0u : 0u);
                    const bool f2 = pk_chroma_edge<false>(c[4], c[5], c[6], c[7], pair_byte(p2.x, p1.x, 2), pair_byte(p2.x, p1.x, 3), pair_byte(tcr4, tcb4, 2) + pk_splat(1), byte_on(bsh, 2), 0u);
                    const uint32_t par = j & 1;
                    if (f0) {
                        if (has_top) LST1(ca + 112 + par, c[1].x), LST1(ca + 114 + par, c[1].y);
                        LST1(co + par, c[2].x), LST1(co + 2 + par, c[2].y);
                    }
                    if (f2) {
                        LST1(co + 48 + par, c[5].x), LST1(co + 50 + par, c[5].y);
                        LST1(co + 64 + par, c[6].x), LST1(co + 66 + par, c[6].y);
                    }
                }
            }
            STAMP(3);
            // ---- 4. what the loads brought: column x + 1 of every sub-row into its window slot (free since this step's stores), the next step's DbPrm into the stage ----
            loads_landed(); // (issued at the top of this step: a step old)
            STAMP(8);
            {
                // (written out once per phase of t + 1 mod 8: in a phase the piece a sub-row is due -- and so the lanes that hold it -- is a constant)
                const int kq = (t + 1) & 7;
                auto piece_l = [&](const v4u &Ga, const v4u &Gb, int k, int ph) {
                    const int c = t - k + 1;
                    if (k < rows_here && c >= 0 && c < wmb && ll_col == ((ph - k) & 7)) LST16(ll_dst + static_cast<uint32_t>(k) * T_BYTES, Ga), LST16(ll_dst + static_cast<uint32_t>(k) * T_BYTES + 128, Gb);
                };
                auto piece_c = [&](const v4u &Gc, const v4u &Gd, int k, int ph) { // column c = t + 1 - k: place in its half line p = c & 7 = (ph - k) & 7, half (c >> 3) & 1
                    const int c = t - k + 1, p = (ph - k) & 7;
                    if (k < rows_here && c >= 0 && c < wmb && lc_q == (p >> 1)) {
                        if (c & 8)
                            LST8(lc_dst + static_cast<uint32_t>(k) * T_BYTES + (p & 3) * 128, (p & 1) ? (v2u{Gd.z, Gd.w}) : (v2u{Gd.x, Gd.y}));
                        else
                            LST8(lc_dst + static_cast<uint32_t>(k) * T_BYTES + (p & 3) * 128, (p & 1) ? (v2u{Gc.z, Gc.w}) : (v2u{Gc.x, Gc.y}));
                    }
                };
#define PIECES(ph)                                                                                                                                          \
    piece_l(GA0, GB0, 0, ph), piece_l(GA1, GB1, 1, ph), piece_l(GA2, GB2, 2, ph), piece_l(GA3, GB3, 3, ph), piece_l(GA4, GB4, 4, ph), piece_l(GA5, GB5, 5, ph), \
        piece_l(GA6, GB6, 6, ph), piece_l(GA7, GB7, 7, ph), piece_c(GC0, GD0, 0, ph), piece_c(GC1, GD1, 1, ph), piece_c(GC2, GD2, 2, ph), piece_c(GC3, GD3, 3, ph),           \
        piece_c(GC4, GD4, 4, ph), piece_c(GC5, GD5, 5, ph), piece_c(GC6, GD6, 6, ph), piece_c(GC7, GD7, 7, ph)
                switch (kq) {
                case 0: PIECES(0); break;
                case 1: PIECES(1); break;
                case 2: PIECES(2); break;
                case 3: PIECES(3); break;
                case 4: PIECES(4); break;
                case 5: PIECES(5); break;
                case 6: PIECES(6); break;
                default: PIECES(7); break;
                }
#undef PIECES
                // (measured: with the lanes' columns rotated by the sub-row, so that all due pieces sit in the same lanes and leave under ONE lane mask -- a fifth of the
                // instructions here --, this phase and the launch take the same time: it is the LDS draining, not the issue)
                STAMP(9);
                if (lane < 40) LST16(stage0 + static_cast<uint32_t>((t + 1) & 1) * (8 * static_cast<uint32_t>(sizeof(DbPrm))) + lane * 16, GP);
            }
            WAVE_SYNC();
            STAMP(6);
        }
#if defined(MI_DB_STATS)
        if (g == 0 && lane_v == 0)
            for (int k = 0; k < 12; k++) atomicAdd(xstatus + 8 + k, st_acc[k]);
        if (blockIdx.x == 0 && lane_v == 0 && g < 15) // when the groups of the launch's first picture ran (100 MHz ticks): the last launch's values stay
            xstatus[32 + 2 * g] = st_t0, xstatus[33 + 2 * g] = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
#endif
    }
}

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
static_assert(MI_DEBLOCK8_WAVE_BYTES == 9 * T_BYTES && T_BYTES >= 1536 && (T_BYTES / 4) % 32 == 8, "LDS layout constants");
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
// Prefetch loads are issued through inline assembly and waited for by ONE explicit s_waitcnt at the end of a step: gfx9 counts loads and
// stores in one in-order counter, and the compiler, which cannot see across the loop's back edge which registers a load may still be
// writing, guards their every use -- a guard behind freshly issued memory operations puts their whole round trip (18 k clocks
// measured) into the step.  In-out operands: lanes the load is predicated off for keep the register's value, and the value only
// ever flows on through the wait's operands, so no compiler-made copy can read a register before its load has landed.
#define ALD16(dst, base, off) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(static_cast<uint32_t>(off)), "s"(base))
#define ALD8(dst, base, off) asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(dst) : "v"(static_cast<uint32_t>(off)), "s"(base))
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
    const v2u z2 = v2u{0u, 0u};
    for (int g = wave; g < ngroups; g += nwaves) {
        int lane = lane_v;
        OPAQUE(lane);
        const int s = lane >> 3, j = lane & 7; // sub-row inside the group, lane inside the macroblock
        const int mby = g * 8 + s;
        const bool row_ok = mby < hmb, has_top = mby > 0;
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
        // where this lane's samples are in HBM: rows 2j, 2j + 1 of the sub-row's macroblocks (input), row pair j - 2 (output: rows -4..11)
        const uint32_t rowmb = static_cast<uint32_t>(row_ok ? mby : 0);
        const uint32_t yin = y_off + (rowmb * 16 + 2 * j) * W;                                         // + W: the second row
        const uint32_t cin = (rowmb * 8 + j) * Wc;                                                     // + cb_off / cr_off
        const bool fl_ok = row_ok && (j >= 2 || has_top), fc_ok = row_ok && (j >= 1 || has_top);      // lanes whose output rows exist
        const uint32_t yout = y_off + (rowmb * 16 + 2 * j - (fl_ok ? 4 : 0)) * W;
        const uint32_t cout = (rowmb * 8 + j - (fc_ok ? 1 : 0)) * Wc;
        // Input registers.  Slot sl of P* / Q* holds macroblock column c with (c + s) % 4 == sl, so that at step t every
        // sub-row consumes slot t % 4 (a wave-uniform register index) although the sub-rows are one column apart.
        v4u PA0 = z4, PA1 = z4, PA2 = z4, PA3 = z4, PB0 = z4, PB1 = z4, PB2 = z4, PB3 = z4; // luma rows 2j, 2j + 1: 4 slots x 16 bytes
        v2u QB0 = z2, QB1 = z2, QB2 = z2, QB3 = z2, QR0 = z2, QR1 = z2, QR2 = z2, QR3 = z2; // chroma row j of Cb, of Cr: 4 slots x 8 bytes
        auto prefetch_group = [&](int gb) { // gb = first macroblock of an aligned group of four
            if (!row_ok || gb >= wmb) return;
            const uint32_t ya = yin + gb * 16, yb = ya + W, ccb = cb_off + cin + gb * 8, ccr = cr_off + cin + gb * 8;
            const int left = wmb - gb;
            const int j0 = (0 - s) & 3, j1 = (1 - s) & 3, j2 = (2 - s) & 3, j3 = (3 - s) & 3; // position inside the group of slot sl
            if (j0 < left) { ALD16(PA0, py, ya + j0 * 16); ALD16(PB0, py, yb + j0 * 16); ALD8(QB0, py, ccb + j0 * 8); ALD8(QR0, py, ccr + j0 * 8); }
            if (j1 < left) { ALD16(PA1, py, ya + j1 * 16); ALD16(PB1, py, yb + j1 * 16); ALD8(QB1, py, ccb + j1 * 8); ALD8(QR1, py, ccr + j1 * 8); }
            if (j2 < left) { ALD16(PA2, py, ya + j2 * 16); ALD16(PB2, py, yb + j2 * 16); ALD8(QB2, py, ccb + j2 * 8); ALD8(QR2, py, ccr + j2 * 8); }
            if (j3 < left) { ALD16(PA3, py, ya + j3 * 16); ALD16(PB3, py, yb + j3 * 16); ALD8(QB3, py, ccb + j3 * 8); ALD8(QR3, py, ccr + j3 * 8); }
        };
        // the macroblock's DbPrm, one step ahead: this lane's strengths (one dword per direction) and the three planes' parameter blocks
        v2u pre_bs = z2;
        v4u pre_p0 = z4, pre_p1 = z4, pre_p2 = z4;
        auto prefetch_prm = [&](int mbx) { // unconditional (a clamped address where there is no such macroblock): a predicated load would make the compiler merge old and new registers with copies
            const uint32_t o = (rowmb * static_cast<uint32_t>(wmb) + static_cast<uint32_t>(min(max(mbx, 0), wmb - 1))) * static_cast<uint32_t>(sizeof(DbPrm));
            ALD8(pre_bs, prms, o + (j >> 1) * 8);
            ALD16(pre_p0, prms, o + 32);
            ALD16(pre_p1, prms, o + 48);
            ALD16(pre_p2, prms, o + 64);
        };
        // every prefetch load issued so far has landed
        auto loads_landed = [&]() {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(PA0), "+v"(PA1), "+v"(PA2), "+v"(PA3), "+v"(PB0), "+v"(PB1), "+v"(PB2), "+v"(PB3));
            asm volatile("" : "+v"(QB0), "+v"(QB1), "+v"(QB2), "+v"(QB3), "+v"(QR0), "+v"(QR1), "+v"(QR2), "+v"(QR3));
            asm volatile("" : "+v"(pre_bs), "+v"(pre_p0), "+v"(pre_p1), "+v"(pre_p2));
        };
        prefetch_group(0);
        prefetch_prm(-s); // step 0 (only sub-row 0 is active)
        loads_landed();
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
        const int nsteps = wmb + 9; // sub-row s: columns in steps s .. s + wmb - 1, the last column's output two steps later
#if defined(MI_DB_STATS)
        uint32_t st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
        for (int t = 0; t < nsteps; t++) {
            STAMP(5); // loop control + whatever the compiler moved across the step boundary
            int lane = lane_v;
            OPAQUE(lane);
            const int s = lane >> 3, j = lane & 7;
            const int mby = g * 8 + s;
            const bool row_ok = mby < hmb, has_top = mby > 0, last_row = mby == hmb - 1;
            const int mbx = t - s;
            const bool active = row_ok && mbx >= 0 && mbx < wmb;
            // LDS addresses: this sub-row's window (index s + 1 of the wavefront's nine), the window above, the slots of columns x and x - 1
            const uint32_t tile = MI_DEBLOCK8_HDR_BYTES + static_cast<uint32_t>(wave) * MI_DEBLOCK8_WAVE_BYTES + static_cast<uint32_t>(s + 1) * T_BYTES;
            const uint32_t sx = static_cast<uint32_t>(mbx) & 3u, spv = static_cast<uint32_t>(mbx - 1) & 3u, sp2 = static_cast<uint32_t>(mbx - 2) & 3u;
            const uint32_t own_l = tile + sx * 256, prev_l = tile + spv * 256, own_c = tile + T_CHROMA + sx * 128, prev_c = tile + T_CHROMA + spv * 128;
            // this step's input registers (wave-uniform slot) and parameters
            // (the unpacking into one register per sample position is written out once per slot behind a wave-uniform switch: selecting the slot's
            // registers with v_cndmask costs three instructions per dword, indexing them dynamically makes the compiler copy all 48 around the loop)
            const int ts = t & 3;
            pk2 v[20];  // luma columns -4..15 of rows 2j | 2j + 1 (vertical pass)
            pk2 cv[10]; // chroma columns -2..7 of Cb | Cr
            auto unpack_l = [&](const v4u &a, const v4u &b) {
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int m = 0; m < 4; m++) v[4 + 4 * k + m] = pair_byte(b[k], a[k], m);
            };
            auto unpack_c = [&](const v2u &cb, const v2u &cr) {
#pragma unroll
                for (int k = 0; k < 2; k++)
#pragma unroll
                    for (int m = 0; m < 4; m++) cv[2 + 4 * k + m] = pair_byte(cr[k], cb[k], m);
            };
            switch (ts) {
            case 0: unpack_l(PA0, PB0), unpack_c(QB0, QR0); break;
            case 1: unpack_l(PA1, PB1), unpack_c(QB1, QR1); break;
            case 2: unpack_l(PA2, PB2), unpack_c(QB2, QR2); break;
            default: unpack_l(PA3, PB3), unpack_c(QB3, QR3); break;
            }
            const v2u bs = pre_bs;
            const v4u p0 = pre_p0, p1 = pre_p1, p2 = pre_p2;
            __builtin_amdgcn_sched_barrier(0); // (everything that reads registers written by vector memory loads stays above this line)
            STAMP(0);
            // ---- 0. the step's vector memory operations in one burst: loads for the steps to come, then the stores of column x - 2; the one wait for
            // them is at the end of the step (loads_landed)
            if (active && (mbx & 3) == 3) prefetch_group(mbx + 1); // the input registers of this sub-row are free again
            prefetch_prm(mbx + 1);
            STAMP(7);
            // column x - 2 leaves for HBM: rows -4..11 (lane j: row pair j - 2; pairs -2, -1 are rows 12..15 of the window above), chroma rows -1..6.
            // (final since the vertical pass of the previous step)
            if (row_ok && mbx >= 2 && mbx <= wmb + 1) {
                const uint32_t above = tile - T_BYTES;
                if (fl_ok) {
                    const uint32_t src = (j < 2 ? above + 192 + j * 32 : tile + (j - 2) * 32) + sp2 * 256;
                    const v4u w0 = LLD16(src), w1 = LLD16(src + 16);
                    const v4u ra = v4u{PERM(w0.y, w0.x, 0x06040200u), PERM(w0.w, w0.z, 0x06040200u), PERM(w1.y, w1.x, 0x06040200u), PERM(w1.w, w1.z, 0x06040200u)};
                    const v4u rb = v4u{PERM(w0.y, w0.x, 0x07050301u), PERM(w0.w, w0.z, 0x07050301u), PERM(w1.y, w1.x, 0x07050301u), PERM(w1.w, w1.z, 0x07050301u)};
                    const uint32_t o = yout + (mbx - 2) * 16;
                    GST16(py, o, ra), GST16(py, o + W, rb);
                }
                if (fc_ok) {
                    const uint32_t src = (j < 1 ? above + T_CHROMA + 112 : tile + T_CHROMA + (j - 1) * 16) + sp2 * 128;
                    const v4u w = LLD16(src);
                    const uint32_t o = cout + (mbx - 2) * 8;
                    GST8(py, cb_off + o, (v2u{PERM(w.y, w.x, 0x05040100u), PERM(w.w, w.z, 0x05040100u)}));
                    GST8(py, cr_off + o, (v2u{PERM(w.y, w.x, 0x07060302u), PERM(w.w, w.z, 0x07060302u)}));
                }
                if (last_row) { // nothing below will touch rows 12..15 (chroma row 7): they leave with the rest
                    if (j < 2) {
                        const uint32_t src = tile + 192 + j * 32 + sp2 * 256;
                        const v4u w0 = LLD16(src), w1 = LLD16(src + 16);
                        const v4u ra = v4u{PERM(w0.y, w0.x, 0x06040200u), PERM(w0.w, w0.z, 0x06040200u), PERM(w1.y, w1.x, 0x06040200u), PERM(w1.w, w1.z, 0x06040200u)};
                        const v4u rb = v4u{PERM(w0.y, w0.x, 0x07050301u), PERM(w0.w, w0.z, 0x07050301u), PERM(w1.y, w1.x, 0x07050301u), PERM(w1.w, w1.z, 0x07050301u)};
                        const uint32_t o = y_off + (static_cast<uint32_t>(mby) * 16 + 12 + 2 * j) * W + (mbx - 2) * 16;
                        GST16(py, o, ra), GST16(py, o + W, rb);
                    } else if (j == 2) {
                        const v4u w = LLD16(tile + T_CHROMA + 112 + sp2 * 128);
                        const uint32_t o = (static_cast<uint32_t>(mby) * 8 + 7) * Wc + (mbx - 2) * 8;
                        GST8(py, cb_off + o, (v2u{PERM(w.y, w.x, 0x05040100u), PERM(w.w, w.z, 0x05040100u)}));
                        GST8(py, cr_off + o, (v2u{PERM(w.y, w.x, 0x07060302u), PERM(w.w, w.z, 0x07060302u)}));
                    }
                }
            }
            STAMP(4);
            // ---- 1. vertical edges: lane j = luma rows 2j, 2j + 1, then chroma row j of Cb | Cr ----
            if (active) {
                // a plane's block: aL bL aI bI | aT bT tL1 tL2 | tL3 tI1 tI2 tI3 | tT1 tT2 tT3 pad  (a / b: alpha / beta of the left-edge, inner, top-edge QP average; tKb: tC0 for bS b).
                // bS -> position of its tC0 byte in {y, z}: one permute maps the four strengths, a second one fetches the four bytes.
                const uint32_t bsv = bs.x;
                const uint32_t tsel = PERM(0x0403020Cu, 0x0706050Cu, bsv + 4u); // edge 0: the left-edge row (bytes 2..4 of {y, z}), inner edges: bytes 5..7; bS 0: a zero
                const bool left = mbx > 0;
                {
                    uint32_t w6 = 0, w7 = 0;
                    bool f0 = false;
                    if (__builtin_amdgcn_ballot_w64(bsv != 0) != 0) {
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
                if (__builtin_amdgcn_ballot_w64(bsc != 0) != 0) {
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
            // ---- 2. hand-off ----
            // 2a. the group's last row: rows 12..15 of column xl - 1 are final but for the row below -- into the ring of the group below
            //     (six 16-byte pieces in window format), then the column is published.  Back-pressure first: the slot held column
            //     xl - 1 - depth, which the group below must have consumed.
            if (feeds_group) {
                const int xl = t - last_sub; // column of the group's last row in this step
                if (xl >= 1 && xl <= wmb) {
                    const int c = xl - 1;
                    if (c >= out_depth) wait_for(&sh.cons[g + 1], c - out_depth + 1);
                    if (s == last_sub && j < 6) {
                        const uint32_t src = j < 4 ? tile + spv * 256 + 192 + j * 16 : tile + T_CHROMA + spv * 128 + 96 + (j - 4) * 16;
                        LST16(out_ring + static_cast<uint32_t>(c % out_depth) * MI_DEBLOCK_SLOT_BYTES + j * 16, LLD16(src));
                    }
                    WAVE_SYNC();
                    if (lane == 0) __hip_atomic_store(&sh.prog[g], xl, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            // 2c. the group's first row takes rows 12..15 of column t of the group above out of its ring, once that says the column is final
            if (g > 0 && t < wmb) {
                wait_for(&sh.prog[g - 1], t + 1);
                if (s == 0 && j < 6) {
                    const uint32_t above = tile - T_BYTES;
                    const uint32_t dst = j < 4 ? above + sx * 256 + 192 + j * 16 : above + T_CHROMA + sx * 128 + 96 + (j - 4) * 16;
                    LST16(dst, LLD16(in_ring + static_cast<uint32_t>(t % in_depth) * MI_DEBLOCK_SLOT_BYTES + j * 16));
                }
            }
            WAVE_SYNC();
            if (g > 0 && t < wmb && lane == 0) // the hand-off slot of column t has been copied: the group above may reuse it
                __hip_atomic_store(&sh.cons[g], t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            STAMP(2);
            // ---- 3. horizontal edges: lane j = luma columns 2j, 2j + 1, then chroma column j of Cb | Cr ----
            if (active) {
                const uint32_t bsh = bs.y;
                const uint32_t above = tile - T_BYTES;
                const uint32_t tsel = PERM(0x0605040Cu, 0x0302010Cu, bsh + 4u); // edge 0: the top-edge row (bytes 0..2 of w), inner edges: bytes 1..3 of z
                if (__builtin_amdgcn_ballot_w64(bsh != 0) != 0) {
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
            WAVE_SYNC();
            STAMP(3);
            loads_landed(); // (issued at the top of this step: a step old)
            STAMP(6);
        }
#if defined(MI_DB_STATS)
        if (g == 0 && lane_v == 0)
            for (int k = 0; k < 12; k++) atomicAdd(xstatus + 8 + k, st_acc[k]);
#endif
    }
}

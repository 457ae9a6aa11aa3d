// h264decode_amd/csrc/k_entropy.hip -- K1/K2: slice_data() entropy decoding on gfx950.
//
// One slice per wavefront (one 64-thread workgroup per slice).  The arithmetic decoder is a
// serial dependency chain, so the syntax-element code is written wave-uniformly: every lane
// executes the same decision sequence, and the lanes fan out only for the wide work --
//   * bitstream: three VGPRs hold 192 consecutive RBSP words, one per lane, and feed a scalar 64-bit look-ahead through
//     v_readlane; the window slides at macroblock boundaries (no LDS ring, no barrier);
//   * context initialisation: the states of the (table set, QP) row of DevTables -- macroblock-level ones one per lane in
//     two VGPRs, residual ones in LDS, gathered into a third VGPR per block category;
//   * neighbour caches: left / row-above state (modes, nnz, refs, mvs, |mvd|) as 48-byte (B: 72-byte) TopInfo entries, the
//     row above in global memory behind a two-entry LDS window;
//   * write-out: the 128-byte MbRec and the coefficient blocks are assembled in LDS; the record leaves with
//     one dword per lane, and only the 16-coefficient blocks that carry anything go to a packed per-pass pool.
// CABAC engine state (codIRange, scaled codIOffset, lookahead count) is wave-uniform and lives on the vector side; Tables
// 9-44 / 9-45 are per-lane tables read with v_readlane (Ent, below).
//
// Code-size discipline: the instruction cache is shared, and hundreds of slices run different parts
// of this kernel at once, so every syntax routine is inlined exactly ONCE: residual blocks, motion
// partitions and reference indices are decoded by single loops over small schedules instead of
// per-case call sites, and block categories are run-time parameters (tables below), not templates.
//
// Replaces: NewSliceData / MbPred (h264/slice.go:570-830, :252-454) and the arithmetic decoding
// engine (h264/cabac.go:439-553); residual parsing, Intra4x4PredMode derivation and motion vector
// prediction are absent from the reference and follow ITU-T H.264 7.3.5, 8.3.1.1, 8.4.1, 9.2, 9.3.
#include <hip/hip_runtime.h>
#include "mi_kernels.h"

// Register budget: 512 / MI_ENT_MINWAVES VGPRs per wavefront.  The reconstruction kernels of the previous pass
// must find free registers next to the long-lived entropy wavefronts: measured at 256 streams, 6 (80 VGPRs, a few
// spills) gives 19.1k frames/s against 18.2k for 4 and 18.4k for 8.
#ifndef MI_ENT_ISLICE_PRIO
#define MI_ENT_ISLICE_PRIO 0
#endif
#ifndef MI_ENT_MINWAVES
#define MI_ENT_MINWAVES 6
#endif
// The file is compiled twice: as k_entropy (I and P slices) and, from k_entropy_b.hip with MI_ENT_B = 1, as k_entropy_b (B
// slices only: two reference lists, direct prediction, Tables 7-14 / 7-18).  Everything list-dependent is indexed by a list
// number that is the constant 0 in the first build, so the I/P kernel carries none of the B machinery.
#ifndef MI_ENT_B
#define MI_ENT_B 0
#endif
// Third build (k_entropy_f.hip: MI_ENT_FMO = 1): the I/P kernel with the slice-group walk (8.2.2) -- launched instead of k_entropy
// for the launches that hold a picture with more than one slice group, so that the common kernel's register budget does not pay
// for a Baseline-only feature.  The B build always carries it.
#ifndef MI_ENT_FMO
#define MI_ENT_FMO MI_ENT_B
#endif
#if MI_ENT_B
#define NL 2
#define MI_ENT_KERNEL k_entropy_b
#elif MI_ENT_FMO
#define NL 1
#define MI_ENT_KERNEL k_entropy_f
#else
#define NL 1
#define MI_ENT_KERNEL k_entropy
#endif
// The lane number as the slice loop sees it: refreshed through an opaque move at the top of every macroblock (Ent::lane), so
// that the compiler recomputes lane predicates (one v_cmp) where they are used instead of hoisting dozens of them out of
// the macroblock loop into scalar register pairs -- which it then spills into VGPR lanes and reloads with two v_readlane
// each.  Every user has an `e` in scope.
#define LANE (e.lane)
#define FI __device__ __forceinline__
// The workgroup is ONE wavefront: cross-lane LDS visibility needs no s_barrier and, above all, no
// wait for outstanding global stores (what __syncthreads() implies) -- LDS operations of a wavefront
// are processed in issue order, so ordering the instructions is enough.
#define LDS_SYNC()                                             \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

struct TopInfo { // edge state of a decoded MB as seen by its right / lower neighbours (48 bytes; 72 in the B build)
    uint8_t type, t8x8, cbp, chroma_mode, cbf_dc;
    uint8_t dmask;  // B: bit k = the k-th 8x8 block on the edge is predicted in direct mode (ref_idx contexts, 9.3.3.1.1.6)
    uint16_t row;   // macroblock row of the entry + 1 (0: never written).  Pictures with slice groups: a slice's previous visit of a column
                    // need not be the row above, so an entry counts as neighbour B / C / D only if its row is the current one - 1
    int8_t ipm[4];  // bottom row (top[]) or right column (left)
    uint8_t nnz[8]; // luma edge [0..3], Cb edge [4..5], Cr edge [6..7]
    int8_t ref[2][2]; // [list][the two 8x8 blocks on the edge] (the I/P build uses list 0 only; [1] is padding there)
    int16_t mv[NL][4][2];
    uint8_t mvd[NL][4][2];
};
#define TOP_DW (static_cast<int>(sizeof(TopInfo) / 4))
static_assert(sizeof(TopInfo) == (MI_ENT_B ? 72 : 48), "TopInfo layout");

struct Shared {
    uint8_t ctx[464];      // home of the residual-block context states (ctxIdx >= 105); see Ent::wk
    uint8_t posmap[4][64]; // CAVLC: scan index -> position: [0] zig-zag 4x4, [1] zig-zag 4x4 of AC index (k+1), [2] zig-zag 8x8, [3] identity
    int16_t coef[MI_COEF_PER_MB];
    MbRec rec;
    // The neighbour entries as ONE array, so that a lane picks its neighbour by index (an LDS offset) and not by a select of
    // pointers (which the compiler turns into generic pointers with null checks): [NB_LEFT] the macroblock to the left,
    // [NB_TL] the top[] entry of column x-1 as it was for the row above, [NB_TOP], [NB_TOP + 1] the LDS window on the
    // row-above state: columns x and x+1 (the row itself lives in HBM)
    TopInfo nb[4];
    // Neighbour caches of the current MB.  6-wide grids: column 0 = left MB, 1..4 = current MB,
    // 5 = right / top-right; row 0 = MB row above, rows 1..4 = current MB.
    int8_t ipm_c[32];      // -2 unavailable, -1 not (yet) an I_NxN block
    uint8_t nnz_c[32];     // 0x80 = unavailable
    uint8_t nnzc_c[2][12]; // chroma 3x3 grids
    int8_t ref_c[NL][32];  // -2 unavailable or not yet decoded, -1 intra / list not used, >= 0 ref_idx (motion final)
    int8_t refi_c[NL][32]; // ref_idx as soon as parsed (CABAC ctxIdxInc of ref_idx; 0 for direct-predicted blocks)
    alignas(4) int16_t mv_c[NL][32][2];
    alignas(2) uint8_t mvd_c[NL][32][2];
    uint16_t parts[16];    // motion partition schedule: bx | by<<2 | (w-1)<<4 | (h-1)<<6 | shape<<8 | B: Pred_L0 / Pred_L1 bits << 11 (0 = direct)
    int8_t refs8[2][4];    // [list][8x8]
    int8_t sub_type[4];
    uint8_t cur_cbf_dc, pad[3];
    int16_t ref_slot[NL][MI_MAX_REFS]; // frame-pool slot per ref_idx of this slice
    uint32_t skip_tmpl[32];            // the MbRec of a P_Skip macroblock as far as it is the same for the whole slice (pskip_fast)
    uint32_t role[64];                 // what each lane does in fill_caches (build_role)
    uint16_t vlc[MI_VLC_N];            // CAVLC slices: the code tables in compact form (mi_types.h: MI_VLC_*), copied in at the start of the slice
    uint8_t coded[64];                 // CABAC neighbourhood, one entry per bit position of parse_residual_cabac's layout: 1 coded, 2 unavailable (fill_caches)
#if MI_ENT_B
    uint32_t col[20];          // ColRec of the co-located macroblock (8.4.1.2.1)
    alignas(4) int16_t dmv[2][16][2]; // direct-predicted sub-macroblocks of a B_8x8 macroblock, until their turn comes (6.4.11.7)
    int8_t dref[2][4];
    int16_t dsf[MI_MAX_REFS];  // temporal direct: DistScaleFactor per refIdxL0
#endif
};
#define GI(bx, by) (((by) + 1) * 6 + (bx) + 1)
enum { NB_LEFT = 0, NB_TL = 1, NB_TOP = 2 };

// Everything the serial syntax code touches per bin lives in registers:
//   * the bit reader is scalar (64-bit MSB-aligned look-ahead in SGPRs) and is fed from three VGPRs that
//     hold 3 x 64 consecutive RBSP words, one per lane, read with v_readlane -- no LDS ring;
//   * the CABAC context states of the macroblock-level syntax elements (ctxIdx 0..104, 399..401) sit
//     one per lane in two VGPRs (ca, cb); the states of the residual block being decoded are gathered
//     from their LDS home into a third VGPR (wk) for the duration of the block;
//   * Tables 9-44 / 9-45, the 8x8 significance maps and the zig-zag scans are per-lane tables too.
// A decision is then ~35 scalar instructions with no memory access at all.
struct Ent {
    int lane;                // see LANE
    Shared *s;
    TopInfo *top; // [wmb] row-above state of this slice, in global memory (read with L1-bypassing loads)
    uint32_t pre_top; // lanes 0..11: prefetched dwords of top[mbx + 2]
    const DevTables *tab;
    const uint32_t *rbsp32;
    const SliceDesc *sd;
    const PicDesc *pd;
    MbRec *mbrec;
    int16_t *coefs;          // coefficient pool of this pass (32-byte blocks)
    uint32_t *pool_head;     // next free block of the pool (device counter, reset per pass)
    uint32_t pool_blocks;    // pool size
    uint32_t coef_cur, coef_end; // this wavefront's chunk of the pool: [coef_cur, coef_end)
    // bit reader
    uint64_t bitbuf;         // next stream bits, MSB first
    int bcnt;                // valid bits in bitbuf (>= 32 between calls)
    uint32_t wpos, wbase, rbsp_words; // next word to fetch / first word of `win`
    uint32_t win, winn, winx; // lane i: RBSP word wbase + i / + 64 + i / + 128 + i, byte-swapped to MSB-first
    // CABAC engine
    uint32_t range, value;   // range: codIRange << avail (renormalisation then only moves avail); value: see the engine below
    int avail;
    uint32_t ca, cb, wk;     // context states, see above; cb lanes 61..63 = ctxIdx 399..401
    int wk_cat;              // ctxBlockCat whose states are in wk (-1: none)
    uint32_t wk_c0;          // cat_word0 of that category
    int wk_home;             // per lane: ctxIdx this lane of wk mirrors
    int wk_valid;            // per lane: 1 = the lane belongs to the category's own contexts (an int, not a bool: a lane predicate kept in a scalar register pair costs
                             // three mask instructions per residual block to carry around the loop)
    uint32_t v_rlps, v_trans; // lane p: rangeTabLPS[p][0..3] / next state after an LPS for valMPS 0
    uint32_t v_maps;         // lane i: sig8x8[i] | last8x8[i] << 8 | zigzag8[i] << 16 | zigzag4[i & 15] << 24
    uint32_t v_pos;          // lane i: where the coefficient of scan index i goes, one byte per position mode: zigzag4[i & 15] | zigzag4[(i + 1) & 15] << 8 (AC blocks: scan index
                             // i is coefficient i + 1) | zigzag8[i] << 16 | i << 24 (field pictures: the field scans)
    uint32_t v_cat0, v_cat1; // lane ctxBlockCat: packed block-category parameters (cat_word0/1)
    uint32_t v_qpc, v_refslot; // lane i: QPc table entry / frame slot of ref_idx i
    uint32_t v_step;         // lane = residual step: step_word()
    uint32_t aw, bw;         // first dword of the left / upper TopInfo (Nb)
    int v_ipm;               // lanes 0..29: Intra4x4/8x8PredMode grid (same layout and codes as Shared::ipm_c)
    int qp, prev_dqp_nz, mbx, mby, cur_type, err;
    uint32_t qpw0, qpw1;     // QP_Y << 16 | QP_C(Cb) << 24 and QP_C(Cr) of the current QP_Y: the record's bytes 2..4 (set_qp)
    int cabac, islice, wmb, hmb;
    int mono;                // chroma_format_idc 0 (h264/sps.go:226-243): no chroma syntax; the reconstruction's chroma planes are 128 by construction
    int cip, t8x8_mode, cqp_off0, cqp_off1, nref;
    uint64_t mb_base;
#if MI_ENT_B
    int nref1, direct_spatial, d8inf, col_short, direct8; // direct8: 8x8 blocks of the current macroblock predicted in direct mode
    const uint32_t *col;     // ColRec array of RefPicList1[0] (nullptr: none)
    uint32_t v_col;          // lanes 0..19: the co-located macroblock's ColRec, fetched at the start of the macroblock
    MbMv1 *mbmv1;            // list-1 vectors of the pass
#endif
#if MI_ENT_STATS
    uint32_t bins;
    uint64_t tacc[4], tmark; // diagnostics: shader clocks in [0] fill_caches [1] macroblock syntax before residual() [2] residual() [3] record write-out
#endif
};
#if MI_ENT_STATS
#define MI_BINS(e) ((e).bins)
#define MI_COUNT_BIN(e) ((e).bins++)
#define MI_TT0(e) ((e).tmark = __builtin_readcyclecounter())
#define MI_TT(e, k) do { const uint64_t now_ = __builtin_readcyclecounter(); (e).tacc[k] += now_ - (e).tmark; (e).tmark = now_; } while (0)
#if MI_ENT_STATS == 4 /* the syntax of a coded macroblock: [0] mb_type and partition schedule [1] ref_idx, mvd, vector prediction [2] intra modes, cbp, mb_qp_delta [3] everything else */
#define MI_T0(e) ((void)0) /* ([3] counts from one macroblock's last stamp to the next one's first: residual, write-out, skip flag, caches) */
#define MI_T(e, k) ((void)0)
#define MI_R0(e) ((void)0)
#define MI_R(e, k) ((void)0)
#define MI_S(e, k) MI_TT(e, k)
#elif MI_ENT_STATS == 2 /* [0] significance map [1] levels [2] rest of residual_block_cabac [3] rest of residual() */
#define MI_T0(e) ((void)0)
#define MI_T(e, k) ((void)0)
#define MI_R0(e) ((void)0)
#define MI_R(e, k) ((void)0)
#define MI_R0(e) MI_TT0(e)
#define MI_R(e, k) MI_TT(e, k)
#else
#define MI_T0(e) MI_TT0(e)
#define MI_T(e, k) MI_TT(e, k)
#define MI_R0(e) ((void)0)
#define MI_R(e, k) ((void)0)
#endif
#else
#define MI_BINS(e) 0u
#define MI_COUNT_BIN(e) ((void)0)
#define MI_T0(e) ((void)0)
#define MI_T(e, k) ((void)0)
#define MI_R0(e) ((void)0)
#define MI_R(e, k) ((void)0)
#endif
#ifndef MI_S
#define MI_S(e, k) ((void)0)
#endif
#define RFL(x) __builtin_amdgcn_readfirstlane(x)
#define RDL(v, i) static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>(i)))
// v_writelane_b32: clang has no builtin for it, the LLVM intrinsic is bound by name
extern "C" __device__ int mi_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// per block category (ctxBlockCat 0..5; Tables 9-34, 9-40, 9-43):
//   word0 = maxNumCoeff | (coded_block_flag base - 64) << 8 | clamp of numDecodAbsLevelGt1 << 16 | position mode << 20
//           (0 zig-zag 4x4, 1 zig-zag 4x4 of index k+1, 2 zig-zag 8x8, 3 identity) | number of last contexts << 24
//   word1 = significant_coeff_flag base | last_significant_coeff_flag base << 10 | coeff_abs_level_minus1 base << 20
FI uint32_t cat_word0(int c) {
    const uint32_t maxnum[6] = {16, 15, 16, 4, 15, 64}, cbf[6] = {85, 89, 93, 97, 101, 64}, lim[6] = {4, 4, 4, 3, 4, 4}, pm[6] = {0, 1, 0, 3, 1, 2};
    const uint32_t nlast[6] = {15, 14, 15, 3, 14, 9};
    c = c < 6 ? c : 0;
    return maxnum[c] | (cbf[c] - 64) << 8 | lim[c] << 16 | pm[c] << 20 | nlast[c] << 24;
}
// (field pictures: the significance contexts of field-coded blocks -- ctxIdxOffset 277 / 338, 8x8 blocks 436 / 451; h264/slice.go:867-872 field_pic_flag)
FI uint32_t cat_word1(int c, int field) {
    const uint32_t sig[6] = {105, 120, 134, 149, 152, 402}, last[6] = {166, 181, 195, 210, 213, 417}, ab[6] = {227, 237, 247, 257, 266, 426};
    const uint32_t sigf[6] = {277, 292, 306, 321, 324, 436}, lastf[6] = {338, 353, 367, 382, 385, 451};
    c = c < 6 ? c : 0;
    return (field ? sigf[c] : sig[c]) | (field ? lastf[c] : last[c]) << 10 | ab[c] << 20;
}

// ------------------------------------------------------------------ bit reader
FI uint32_t load_win(const Ent &e, uint32_t base) {
    const uint32_t i = base + LANE;
    return i < e.rbsp_words ? __builtin_bswap32(e.rbsp32[i]) : 0u;
}
// The three window VGPRs cover 192 consecutive words from wbase.  The window slides at macroblock
// boundaries only (slide_window): 128 words then remain ahead of the cursor, more than the 3200 bits
// a macroblock_layer() may occupy (A.3.1); I_PCM samples are reached with seek().
FI uint32_t fetch_word(Ent &e) {
    const uint32_t idx = e.wpos - e.wbase;
    e.wpos++;
    const uint32_t w0 = RDL(e.win, idx), w1 = RDL(e.winn, idx), w2 = RDL(e.winx, idx); // the lane select is idx & 63
    return idx < 64 ? w0 : (idx < 128 ? w1 : w2);
}
FI void slide_window(Ent &e) {
    while (e.wpos - e.wbase >= 64) { // twice after a macroblock of more than 2048 bits
        e.win = e.winn, e.winn = e.winx;
        e.wbase += 64;
        e.winx = load_win(e, e.wbase + 128); // consumed 64 words (>= one macroblock) later
    }
}
FI uint32_t bitpos(const Ent &e) { return e.wpos * 32 - static_cast<uint32_t>(e.bcnt); }
FI void seek(Ent &e, uint32_t pos) {
    const uint32_t w = pos >> 5;
    if (w - e.wbase >= 64) {
        e.wbase = w;
        e.win = load_win(e, w), e.winn = load_win(e, w + 64), e.winx = load_win(e, w + 128);
    }
    e.wpos = w;
    const uint64_t hi = fetch_word(e), lo = fetch_word(e);
    e.bitbuf = ((hi << 32) | lo) << (pos & 31);
    e.bcnt = 64 - static_cast<int>(pos & 31);
}
FI uint32_t peek32(const Ent &e) { return static_cast<uint32_t>(e.bitbuf >> 32); }
FI void skip(Ent &e, int n) { // 0..32
    e.bitbuf <<= n;
    e.bcnt -= n;
    if (__builtin_expect(e.bcnt < 32, 0)) {
        const uint64_t w = fetch_word(e);
        e.bitbuf |= w << (32 - e.bcnt);
        e.bcnt += 32;
    }
}
FI uint32_t get_bits(Ent &e, int n) { // 1..32
    const uint32_t v = peek32(e) >> (32 - n);
    skip(e, n);
    return v;
}
FI uint32_t get_bit(Ent &e) { return get_bits(e, 1); }
FI uint32_t get_ue(Ent &e) { // 9.1 with one CLZ
    const uint32_t w = peek32(e);
    if (w == 0) {
        e.err = 1;
        skip(e, 32);
        return 0;
    }
    const int lz = __clz(w);
    if (lz > 15) {
        skip(e, lz + 1);
        return (1u << lz) - 1 + get_bits(e, lz);
    }
    skip(e, 2 * lz + 1);
    return (w >> (31 - 2 * lz)) - 1;
}
FI int get_se(Ent &e) {
    uint32_t k = get_ue(e);
    int m = static_cast<int>((k + 1) >> 1);
    return (k & 1) ? m : -m;
}

// ------------------------------------------------------------------ CABAC engine (9.3.1.2, 9.3.3.2)
// codIOffset is kept scaled: value = (codIOffset << avail) | next `avail` stream bits, and so is codIRange: range = codIRange << avail.
//
// Issue balance.  A compute unit has ONE scalar ALU (about one SALU instruction per cycle for all of
// its wavefronts) next to four vector ALUs, and tens of slices share a CU, so a decoder written purely
// in scalar instructions is bound by that single unit.  The engine is therefore split: the data path
// (codIRange / codIOffset arithmetic, renormalisation, state selection) runs on the VALU with the
// same value in every lane, while table indices, loop control and branches stay scalar.  VGPR() pins
// a wave-uniform value to the vector side; UNI() turns a vector-side condition into a scalar branch
// condition (v_cmp writes the lane mask, one s_cmp tests it).
// Round 4 re-measured the balance (PMC + variant builds): at 256 streams a SIMD's vector pipe is busy ~90 % of the kernel and the P slices are
// bound by it, the I slices by their own dependency chain -- which is why handing the successor-state selection and the bin to the scalar ALU
// (7 vector instructions less per bin, but a vector -> scalar hop behind the compare) lost 3 - 14 %, and why every vector instruction saved
// without such a hop shows up in the step time.
#define VGPR(x) asm volatile("" : "+v"(x))
#define OPAQUE(x) asm volatile("" : "+v"(x))
#define UNI(cond) (__builtin_amdgcn_ballot_w64(cond) != 0)
// a context state as it sits in a register lane: pStateIdx | valMPS << 6 (the byte kept in LDS and in DevTables::ctx_init) and valMPS once more in bit 31
FI uint32_t ctx_word(uint32_t b) { return b | (b & 64u) << 25; }
// 16 more stream bits when fewer than 7 are left below codIOffset (avail < 7, i.e. the scaled range is below 2^15), and the scale of the range as
// it stands now: avail = 23 - clz(range) (codIRange has its bit 8 set after RenormD).  Written this way -- test on the range, avail afterwards -- the
// common path is: compare, branch not taken, count leading zeros, subtract.
FI void cabac_refill(Ent &e) {
    if (__builtin_expect(UNI(e.range < 0x8000u), 0)) { // about once per 13 decisions: keep the common path fall-through
        e.value = (e.value << 16) | (peek32(e) >> 16);
        skip(e, 16);
        e.range <<= 16;
    }
    e.avail = 23 - __builtin_clz(e.range);
}
FI void cabac_start(Ent &e) { // initDecodingEngine, h264/cabac.go:439-446
    e.range = 510;
    e.value = get_bits(e, 9);
    VGPR(e.range);
    VGPR(e.value);
    cabac_refill(e);
    VGPR(e.avail);
}
// DecodeDecision (h264/cabac.go:521-540) + state transition (:544-553) + RenormD (:503-511) on the
// context state held in lane `idx_` of `reg`.  The bin comes back on the vector side (same value in every
// lane): BIN_x() turns it into a branch condition (v_cmp + s_cbranch_vcc), BINI_x() into a scalar integer.
// SM: the lane of the state is selected by a mask made on the scalar side (1 << idx) instead of a lane compare -- for a context index that is not
// a compile-time constant the shift fills one of the wait states between the two v_readlane and saves the v_cmp.  Used in the residual loops
// (196.7 against 200.6 ms); at the macroblock-level sites with computed indices (skip flag, cbp, mvd, coded_block_flag) it measured 1 ms slower.
template <bool SM = false>
FI uint32_t cabac_decide(Ent &e, uint32_t &reg, int idx_) {
    MI_COUNT_BIN(e);
    const int idx = RFL(idx_);
    const uint64_t lanemask = 1ull << (idx & 63);
    // A context state is pStateIdx | valMPS << 6 | valMPS << 31: it selects its own table lane as it stands (v_readlane takes the select modulo
    // 64), so the chain state -> table entries has no scalar instruction in it.
    const uint32_t st = RDL(reg, idx);
    const uint32_t rl4 = RDL(e.v_rlps, st), tr = RDL(e.v_trans, st);
    // everything else runs on the vector side, same value in every lane: st is pinned there
    uint32_t vst = st;
    VGPR(vst);
    // the two candidate successor states.  v_trans: pStateIdx ^ (LPS successor), with bits 6 and 31 set where the MPS flips (pStateIdx 0) -- XOR
    // with the state gives the LPS successor -- and, one byte up, what the MPS path adds to the state (1; 0 at pStateIdx 62): one SDWA add.
    // (Bits 8..30 of a state register may hold leftovers of the table word; nothing reads them.)
    uint32_t next_mps;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(next_mps) : "s"(tr), "v"(vst));
    const uint32_t next_lps = tr ^ vst;
    // e.range is codIRange << avail.  rangeTabLPS[pStateIdx][(codIRange >> 6) & 3]: codIRange >> 6 is 4..7, which as a v_perm selector picks
    // byte 0..3 of the first operand
    const uint32_t rlps = __builtin_amdgcn_perm(rl4, 0u, e.range >> (e.avail + 6)) << e.avail;
    const uint32_t rmps = e.range - rlps;
    const bool lps = e.value >= rmps;
    const uint32_t diff = e.value - rmps; // wraps when value < rmps; both are below 2^31
    e.value = min(e.value, diff);
    e.range = lps ? rlps : rmps;
    if (SM) {
        const uint32_t next = lps ? next_lps : next_mps;
        asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(reg) : "v"(next), "s"(lanemask));
    } else
        reg = LANE == idx ? (lps ? next_lps : next_mps) : reg;
    cabac_refill(e); // RenormD: the scaled range stays as it is, only the scale moves
    return vst ^ diff; // the bin is the complement of the sign: valMPS on the MPS path (diff negative), !valMPS otherwise
}
// a bin as a branch condition (v_cmp + s_cbranch_vcc) / as a scalar integer
#define BIN_A(e, ctx) UNI(static_cast<int>(cabac_decide(e, (e).ca, (ctx))) >= 0)          /* ctxIdx 0..63 */
#define BIN_B(e, ctx) UNI(static_cast<int>(cabac_decide(e, (e).cb, (ctx) - 64)) >= 0)     /* ctxIdx 64..124 */
#define BIN_T8(e, inc) UNI(static_cast<int>(cabac_decide(e, (e).cb, 61 + (inc))) >= 0)    /* ctxIdx 399..401 */
#define BIN_W(e, lane) UNI(static_cast<int>(cabac_decide<true>(e, (e).wk, (lane))) >= 0)        /* residual working set */
#define BINI_A(e, ctx) static_cast<int>(RFL(~cabac_decide(e, (e).ca, (ctx)) >> 31))
#define BINI_B(e, ctx) static_cast<int>(RFL(~cabac_decide(e, (e).cb, (ctx) - 64) >> 31))
#define BINI_T8(e, inc) static_cast<int>(RFL(~cabac_decide(e, (e).cb, 61 + (inc)) >> 31))
FI bool cabac_bypass(Ent &e) { // 9.3.3.2.3 (A9)
    MI_COUNT_BIN(e);
    e.avail -= 1;
    e.range >>= 1; // (codIRange << avail, with avail one less)
    const bool one = e.value >= e.range;
    e.value = min(e.value, e.value - e.range);
    if (__builtin_expect(UNI(e.range < 0x8000u), 0)) {
        e.value = (e.value << 16) | (peek32(e) >> 16);
        skip(e, 16);
        e.range <<= 16;
        e.avail += 16;
    }
    return UNI(one);
}
FI bool cabac_terminate(Ent &e) { // 9.3.3.2.4
    e.range -= 2u << e.avail;
    if (UNI(e.value >= e.range)) return true;
    cabac_refill(e);
    return false;
}
// Exp-Golomb suffix of UEGk binarisations (9.3.2.3), bypass coded
FI int cabac_egk(Ent &e, int k) {
    int v = 0;
    while (cabac_bypass(e)) {
        v += 1 << k;
        if (++k > 24) {
            e.err = 4;
            break;
        }
    }
    while (k--) v += cabac_bypass(e) << k;
    return v;
}

FI void set_qp(Ent &e, int qp) { // QP_Y and the two chroma QPs it maps to (8.5.8: Table 8-15 on qP_I = Clip3(0, 51, QP_Y + chroma_qp_index_offset))
    e.qp = qp;
    e.qpw0 = static_cast<uint32_t>(qp) << 16 | RDL(e.v_qpc, min(max(qp + e.cqp_off0, 0), 51)) << 24;
    e.qpw1 = RDL(e.v_qpc, min(max(qp + e.cqp_off1, 0), 51));
}
// ------------------------------------------------------------------ neighbour MBs
FI bool nb_ok(const Ent &e, int i) { return e.s->nb[i].type != MBT_NONE && (i != NB_TOP + 1 || e.mbx + 1 < e.wmb); } // macroblock D, A, B, C available
// the left / upper macroblock's type, transform flag, cbp and chroma mode as one scalar word (first dword of TopInfo)
struct Nb {
    uint32_t w;
    FI bool ok() const { return (w & 255) != 0; }
    FI int type() const { return static_cast<int>(w & 255); }
    FI int t8x8() const { return static_cast<int>((w >> 8) & 255); }
    FI int cbp() const { return static_cast<int>((w >> 16) & 255); }
    FI int chroma_mode() const { return static_cast<int>(w >> 24); }
};
// L1-bypassing dword load of the row-above array (it is rewritten by this wave one row later)
FI uint32_t top_load(const Ent &e, int col, int dw) {
    return col < e.wmb ? __hip_atomic_load(reinterpret_cast<const uint32_t *>(e.top + col) + dw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
}

// ------------------------------------------------------------------ residual blocks
// residual_block_cabac 7.3.5.3.3 for ctxBlockCat `cat`; coefficients are written de-zig-zagged.
// The block's context states are gathered into e.wk (lanes 0..15 significant_coeff_flag, 16..31
// last_significant_coeff_flag, 32..41 coeff_abs_level_minus1) while coded_block_flag is decoded and
// scattered back to their LDS home afterwards.  Levels are collected in a VGPR (lane = scan index)
// and stored with one predicated LDS write per block.
FI int cabac_residual(Ent &e, int16_t *dst, int cat_, int cbf_inc) {
    MI_R(e, 3);
    const int cat = RFL(cat_);
    int l = LANE;
    OPAQUE(l); // keeps lane-dependent addresses from being hoisted out of the macroblock loop and spilled
    // The working set stays in e.wk from block to block (and macroblock to macroblock) as long as the category does
    // not change -- a macroblock has at most four category runs -- so states move between LDS and the VGPR only then.
    if (cat != e.wk_cat) {
        if (e.wk_valid) e.s->ctx[e.wk_home] = static_cast<uint8_t>(e.wk); // masked scatter of the previous category
        const uint32_t c0n = RDL(e.v_cat0, cat), c1n = RDL(e.v_cat1, cat);
        const int grp = l >> 4, li = l & 15;
        const int home = static_cast<int>((c1n >> (10 * (grp > 2 ? 2 : grp))) & 1023) + (l < 32 ? li : l - 32);
        // lanes outside the category's own context ranges would hold copies of other categories' states and must never
        // be written back (for 8x8 blocks ctxIdx 417 appears in both the sig and the last group)
        const int nsig = cat == 5 ? 15 : static_cast<int>(c0n & 255) - 1, nlast = static_cast<int>((c0n >> 24) & 15);
        e.wk_valid = (grp == 0 ? li < nsig : (grp == 1 ? li < nlast : l < 42)) ? 1 : 0;
        e.wk_home = home < 464 ? home : 463;
        LDS_SYNC(); // the scatter above may alias the gather below
        e.wk = ctx_word(e.s->ctx[e.wk_home]);
        e.wk_cat = cat, e.wk_c0 = c0n;
    }
    const uint32_t c0 = e.wk_c0;
    if (cat != 5 && !BIN_B(e, 64 + ((c0 >> 8) & 255) + cbf_inc)) {
        MI_R(e, 2);
        return 0;
    }
    const int last = static_cast<int>(c0 & 255) - 1; // maxNumCoeff - 1
    const bool is8 = cat == 5;
    // the significant coefficients as a bit set with the HIGHEST frequency in the LOWEST bit (coefficient i = bit 63 - i): the levels come highest
    // frequency first (7.3.5.3.3), which is then find-first-one / clear-that-bit
    uint64_t sig = 0;
    int i;
    MI_R(e, 2);
    // significance map; a set last_significant_coeff_flag ends the loop through the index itself (a jump out of the loop costs the compiler's
    // structurizer more than the select), running off the end means the final coefficient is significant by inference.
    // (ctxIdxInc of a 4x4 / 2x2 block is the scan position itself: Min(numDecod / NumC8x8, 2) only bites with 4:2:2 chroma DC blocks.)
    if (is8) {
        for (i = 0; i < last; i++) {
            const uint32_t m = RDL(e.v_maps, i);
            if (BIN_W(e, m & 255)) {
                sig |= 0x8000000000000000ull >> i;
                if (BIN_W(e, 16 + ((m >> 8) & 255))) i = 64;
            }
        }
    } else {
        for (i = 0; i < last; i++) {
            if (BIN_W(e, i)) {
                sig |= 0x8000000000000000ull >> i;
                if (BIN_W(e, 16 + i)) i = 64;
            }
        }
    }
    if (i == last) sig |= 0x8000000000000000ull >> last;
    MI_R(e, 0);
    const int n = __builtin_popcountll(sig);
    // levels, highest frequency first (9.3.3.1.3): inc0 / cx are the wk lanes of the two context selections
    int inc0 = 33, cx = 37;
    const int cxmax = 37 + static_cast<int>((c0 >> 16) & 15);
    int lv = 0;
    const int lrev = 63 - l;
    while (sig) {
        const int k = __builtin_ctzll(sig);
        asm("s_bitset0_b64 %0, %1" : "+s"(sig) : "s"(k));
        int a = 1;
        if (BIN_W(e, inc0)) {
            a = 2;
            while (a < 15 && BIN_W(e, cx)) a++;
            if (a >= 15) a += cabac_egk(e, 0);
            inc0 = 32;
            cx = cx < cxmax ? cx + 1 : cxmax;
        } else if (inc0 != 32)
            inc0 = inc0 < 36 ? inc0 + 1 : 36;
        const int v = cabac_bypass(e) ? -a : a;
        lv = lrev == k ? v : lv;
    }
    MI_R(e, 1);
    {
        const uint32_t pos = __builtin_amdgcn_ubfe(e.v_pos, (c0 >> 17) & 0x18u, 8); // the byte of the category's position mode: one bit-field extract, no branch
        if (lv != 0) dst[pos] = static_cast<int16_t>(lv);
    }
    MI_R(e, 2);
    return n;
}

// residual_block_cavlc 9.2.  kind: 0 = 16 coefficients, 1 = 15 (AC), 2 = chroma DC (4),
// 3 = 16 coefficients that are every fourth one of an 8x8 block's scan (CAVLC + 8x8 transform interleave: scan index pmul * i + padd)
FI int cavlc_residual(Ent &e, int16_t *dst, int kind, int nC, int pmul, int padd) {
    MI_R(e, 3); // (-DMI_ENT_STATS=2 on a CAVLC slice: [0] coeff_token [1] levels [2] total_zeros, run_before, store [3] rest of residual())
    const int maxnum = kind == 1 ? 15 : (kind == 2 ? 4 : 16);
    uint32_t w = peek32(e);
    const int wlz = w ? __clz(w) : 32;
    uint32_t ent;
    // coeff_token: compact tables in LDS, entry chosen by the leading zeros and the bits behind the first one (mi_types.h)
    if (kind == 2)
        ent = e.s->vlc[MI_VLC_CDC + MI_VLC_INDEX(w, wlz, MI_VLC_CDC_L, MI_VLC_CDC_S)];
    else if (nC < 2)
        ent = e.s->vlc[MI_VLC_CT0 + MI_VLC_INDEX(w, wlz, MI_VLC_CT0_L, MI_VLC_CT_S)];
    else if (nC < 4)
        ent = e.s->vlc[MI_VLC_CT1 + MI_VLC_INDEX(w, wlz, MI_VLC_CT1_L, MI_VLC_CT_S)];
    else if (nC < 8)
        ent = e.s->vlc[MI_VLC_CT2 + MI_VLC_INDEX(w, wlz, MI_VLC_CT2_L, MI_VLC_CT_S)];
    else
        ent = e.s->vlc[MI_VLC_CT3 + (w >> 26)];
    ent = RFL(ent);
    if (!(ent >> 8)) {
        e.err = 6;
        return 0;
    }
    skip(e, ent >> 8);
    const int total = (ent >> 2) & 31, t1s = ent & 3;
    MI_R(e, 0);
    if (total == 0) return 0;
    if (total > maxnum) {
        e.err = 7;
        return 0;
    }
    int suffix_len = (total > 10 && t1s < 3) ? 1 : 0;
    // levels and scan positions are collected in registers, lane i = the i-th coefficient (highest frequency first), and leave with ONE predicated
    // LDS store at the end -- no scratch array written and read back coefficient by coefficient
    int l = LANE;
    OPAQUE(l);
    int lvv = 0, posv = 0;
    for (int i = 0; i < total; i++) {
        int lv;
        if (i < t1s)
            lv = 1 - 2 * static_cast<int>(get_bit(e));
        else {
            uint32_t ww = peek32(e);
            if (ww == 0) {
                e.err = 8;
                return 0;
            }
            int prefix = __clz(ww);
            int code = (prefix < 15 ? prefix : 15) << suffix_len;
            if (prefix < 14) { // the common case: prefix, stop bit and suffix come out of the one 32-bit look-ahead (at most 14 + 1 + 6 bits)
                if (suffix_len > 0) code += static_cast<int>((ww << (prefix + 1)) >> (32 - suffix_len));
                skip(e, prefix + 1 + suffix_len);
            } else {
                skip(e, prefix + 1);
                int size = (prefix == 14 && suffix_len == 0) ? 4 : (prefix >= 15 ? prefix - 3 : suffix_len);
                if (size > 0) code += get_bits(e, size);
            }
            if (prefix >= 15 && suffix_len == 0) code += 15;
            if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
            if (i == t1s && t1s < 3) code += 2;
            lv = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
            if (suffix_len == 0) suffix_len = 1;
            int al = lv < 0 ? -lv : lv;
            if (al > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
        }
        lvv = l == i ? lv : lvv;
    }
    MI_R(e, 1);
    int zeros_left = 0;
    if (total < maxnum) {
        uint32_t ww = peek32(e);
        const int zlz = ww ? __clz(ww) : 32;
        const uint32_t en = RFL(static_cast<uint32_t>(kind == 2 ? e.s->vlc[MI_VLC_CDCTZ + 8 * (total - 1) + (ww >> 29)]
                                                                : e.s->vlc[MI_VLC_TZ + (total - 1) * MI_VLC_TZ_STRIDE + MI_VLC_INDEX(ww, zlz, MI_VLC_TZ_L, MI_VLC_TZ_S)]));
        if (!(en >> 8)) {
            e.err = 9;
            return 0;
        }
        skip(e, en >> 8);
        zeros_left = en & 255;
    }
    int pos = zeros_left + total - 1; // scan position of the highest-frequency coefficient
    if (pos >= maxnum) {
        e.err = 10;
        return 0;
    }
    const uint8_t *pm = e.s->posmap[kind == 0 ? 0 : (kind == 1 ? 1 : (kind == 3 ? 2 : 3))]; // (kind 3: scan index i of this 4x4 part is index 4 i + part of the 8x8 scan)
    for (int i = 0; i < total; i++) {
        posv = l == i ? pos : posv;
        if (i < total - 1) {
            int run = 0;
            if (zeros_left > 0) {
                uint32_t ww = peek32(e);
                const int rlz = ww ? __clz(ww) : 32;
                const uint32_t en = zeros_left > 6 ? MI_RUN_BEFORE_LONG(ww, rlz) : RDL(e.v_cat0, 8 * (zeros_left - 1) + (ww >> 29)); // (CAVLC slices: v_cat0 = the run_before tables)
                if (!(en >> 8) || static_cast<int>(en & 255) > zeros_left) {
                    e.err = 11;
                    return 0;
                }
                skip(e, en >> 8);
                run = en & 255;
                zeros_left -= run;
            }
            pos -= run + 1;
        }
    }
    if (l < total) dst[pm[posv * pmul + padd]] = static_cast<int16_t>(lvv);
    MI_R(e, 2);
    return total;
}

FI int nc_of(uint8_t a, uint8_t b) { // 9.2.1
    int av = !(a & 0x80), bv = !(b & 0x80);
    if (av && bv) return (a + b + 1) >> 1;
    return av ? a : (bv ? b : 0);
}
FI int cbf_inc_of(const Ent &e, uint8_t a, uint8_t b) { // 9.3.3.1.1.9
    int ci = MB_IS_INTRA(e.cur_type);
    int ca = (a & 0x80) ? ci : (a != 0), cb = (b & 0x80) ? ci : (b != 0);
    return ca + 2 * cb;
}

// residual() 7.3.5.3 as ONE loop over a block schedule:
//   0: Intra16x16 DC | 1..16: luma blocks (z-order) | 17,18: chroma DC | 19..26: chroma AC
// residual() 7.3.5.3 for CABAC.  coded_block_flag contexts (9.3.3.1.1.9) need one bit per neighbouring
// block, so the neighbourhood is a 64-bit scalar mask instead of LDS arrays:
//   bits  0..29  luma 4x4 blocks, 6-wide grid GI(bx, by) (column 0 = left MB, row 0 = MB above)
//   bits 32..49  chroma AC blocks, two 3x3 grids (32 + 9 * plane + 3 * (by + 1) + bx + 1)
//   bits 50..52 / 53..55  DC flags (Intra16x16 luma, Cb, Cr) of the left / upper macroblock, 56..58 of this one
// fill_caches() leaves "coded" (1) / "unavailable" (2) per bit position in Shared::coded; unavailable counts as coded for intra
// macroblocks.  The schedule is a bit set of steps (0 Intra16x16 DC, 1..16 luma z-order, 17/18 chroma DC,
// 19..26 chroma AC) and a per-lane descriptor table: A bit | B bit << 6 | own bit << 12 | dst / 4 << 18 (10 bits: the scalar ALU gets the destination and
// the block category out of it with one bit-field extract each) | kind << 26 (bits 24, 25 stay 0: (word >> 24) & 12 is 4 * kind).
FI uint32_t step_word(int st) {
    if (st == 0) return 50u | 53u << 6 | 56u << 12 | (MI_COEF_I16DC / 4) << 18 | 0u << 26;
    if (st <= 16) {
        const int idx = st - 1, bx = (idx & 1) + 2 * ((idx >> 2) & 1), by = ((idx >> 1) & 1) + 2 * (idx >> 3);
        return static_cast<uint32_t>(GI(bx - 1, by)) | static_cast<uint32_t>(GI(bx, by - 1)) << 6 | static_cast<uint32_t>(GI(bx, by)) << 12 |
               static_cast<uint32_t>((by * 4 + bx) * 4) << 18 | 1u << 26;
    }
    if (st <= 18) {
        const uint32_t c = st - 17;
        return (51u + c) | (54u + c) << 6 | (57u + c) << 12 | (MI_COEF_CDC / 4 + c) << 18 | 2u << 26;
    }
    if (st <= 26) {
        const int j = st - 19, c = j >> 2, bx = j & 1, by = (j >> 1) & 1, g = 32 + 9 * c;
        return static_cast<uint32_t>(g + (by + 1) * 3 + bx) | static_cast<uint32_t>(g + by * 3 + bx + 1) << 6 | static_cast<uint32_t>(g + (by + 1) * 3 + bx + 1) << 12 |
               static_cast<uint32_t>(MI_COEF_CAC / 4 + j * 4) << 18 | 3u << 26;
    }
    return 0;
}
FI void parse_residual_cabac(Ent &e, int cbp_luma, int cbp_chroma, int t8x8) {
    Shared *s = e.s;
    const int i16 = e.cur_type == MBT_I16x16;
    // the neighbourhood as one flag per LANE (lane = bit position of the layout above): a block's two context bits are two v_readlane with the
    // step word as lane select, a coded block sets its lane -- the scalar ALU, the scarce unit, does none of it
    int nl = LANE;
    OPAQUE(nl);
    const uint32_t cd = s->coded[nl];
    uint32_t vnz = (cd == 1 || (cd == 2 && MB_IS_INTRA(e.cur_type))) ? 1u : 0u;
    // luma steps of the coded 8x8 blocks: all four 4x4 blocks, or the first one standing for the 8x8 block
    const uint32_t spread = (cbp_luma & 1) | (cbp_luma & 2) << 3 | (cbp_luma & 4) << 6 | (cbp_luma & 8) << 9;
    uint32_t steps = static_cast<uint32_t>(i16) | (spread * (t8x8 ? 1u : 15u)) << 1;
    if (cbp_chroma) steps |= 3u << 17;
    if (cbp_chroma & 2) steps |= 0xFFu << 19;
    const uint32_t cats = 0u | (t8x8 ? 5u : (i16 ? 1u : 2u)) << 4 | 3u << 8 | 4u << 12; // ctxBlockCat by step kind
    while (steps) {
        const int step = __builtin_ctz(steps);
        steps &= steps - 1;
        const uint32_t d = RDL(e.v_step, step);
        const int cat = static_cast<int>((cats >> ((d >> 24) & 12u)) & 15);
        const int own = static_cast<int>((d >> 12) & 63);
        const int dst = cat == 5 ? (step - 1) * 16 : static_cast<int>((d >> 18) & 255) * 4;
        const int inc = static_cast<int>(RDL(vnz, d) + 2 * RDL(vnz, d >> 6)); // (the lane select is taken modulo 64)
        slide_window(e); // (see cavlc: a macroblock_layer() beyond A.3.1's 3200 bits must not run off the window; one block is at most 92 words)
        if (cabac_residual(e, s->coef + dst, cat, inc)) {
            // the block's lane -- an 8x8 block stands for its four 4x4 positions: own, own + 1, own + 6, own + 7 -- through a mask made on the scalar side
            const uint64_t m = (cat == 5 ? 0xC3ull : 1ull) << own;
            asm("v_cndmask_b32_e64 %0, %0, 1, %1" : "+v"(vnz) : "s"(m));
        }
    }
    const uint64_t nzm = __builtin_amdgcn_ballot_w64(vnz != 0);
    // results: deblocking mask (raster 4x4), DC flags and the 0/1 "coded" grids the neighbours will read
    const uint32_t lo = static_cast<uint32_t>(nzm);
    const uint32_t nzmask = ((lo >> GI(0, 0)) & 15) | ((lo >> GI(0, 1)) & 15) << 4 | ((lo >> GI(0, 2)) & 15) << 8 | ((lo >> GI(0, 3)) & 15) << 12;
    s->rec.nzmask = static_cast<uint16_t>(nzmask);
    s->cur_cbf_dc = static_cast<uint8_t>((nzm >> 56) & 7);
    int l = LANE;
    OPAQUE(l); // keeps lane-dependent addresses from being hoisted out of the macroblock loop and spilled
    const int bit = static_cast<int>(vnz);
    if (l < 30) {
        const int gx = l % 6 - 1, gy = l / 6 - 1;
        if (gx >= 0 && gx < 4 && gy >= 0) s->nnz_c[l] = static_cast<uint8_t>(bit);
    } else if (l >= 32 && l < 50) {
        const int i = l - 32, g = i % 9, gx = g % 3 - 1, gy = g / 3 - 1;
        if (gx >= 0 && gy >= 0) s->nnzc_c[i / 9][g] = static_cast<uint8_t>(bit);
    }
}

FI void parse_residual_cavlc(Ent &e, int cbp_luma, int cbp_chroma, int t8x8) {
    Shared *s = e.s;
    const int i16 = e.cur_type == MBT_I16x16;
    int nzmask = 0, dc = 0;
    // total_coeff of the neighbourhood, one count per LANE while the macroblock's blocks are parsed (lanes 0..29: the luma grid of Shared::nnz_c,
    // lanes 32..55: the two chroma grids of Shared::nnzc_c): a block's nA / nB are two v_readlane, its own count one select -- no LDS round trip
    // per block; the grids go back to LDS once, at the end
    int nl = LANE;
    OPAQUE(nl);
    uint32_t vnn = nl < 30 ? s->nnz_c[nl] : ((nl >= 32 && nl < 56) ? (&s->nnzc_c[0][0])[nl - 32] : 0u);
    for (int step = 0; step < 27; step++) {
        int kind, bx = 0, by = 0, la = 0, lb = 0, own = 63, pmul = 1, padd = 0;
        int16_t *dst;
        if (step == 0) {
            if (!i16) continue;
            kind = 0;
            dst = s->coef + MI_COEF_I16DC;
            la = GI(-1, 0), lb = GI(0, -1);
        } else if (step <= 16) {
            const int idx = step - 1, b8 = idx >> 2;
            if (!((cbp_luma >> b8) & 1)) {
                step += 3; // whole 8x8 uncoded
                continue;
            }
            bx = (idx & 1) + 2 * ((idx >> 2) & 1), by = ((idx >> 1) & 1) + 2 * (idx >> 3);
            la = GI(bx - 1, by), lb = GI(bx, by - 1), own = GI(bx, by);
            if (t8x8) { // CAVLC + 8x8 transform: 4x4 "block" b4 carries coefficients 4 * i + b4 of the 8x8 scan (7.3.5.3.2): they go straight to their places
                kind = 3;
                dst = s->coef + b8 * 64;
                pmul = 4, padd = idx & 3;
            } else {
                kind = i16 ? 1 : 0;
                dst = s->coef + (by * 4 + bx) * 16;
            }
        } else if (step <= 18) {
            if (!cbp_chroma) break;
            const int c = step - 17;
            kind = 2;
            dst = s->coef + MI_COEF_CDC + 4 * c;
        } else {
            if (!(cbp_chroma & 2)) break;
            const int j = step - 19, c = j >> 2, b4 = j & 3;
            bx = b4 & 1, by = b4 >> 1;
            kind = 1;
            dst = s->coef + MI_COEF_CAC + j * 16;
            la = 32 + 12 * c + (by + 1) * 3 + bx, lb = 32 + 12 * c + by * 3 + bx + 1, own = 32 + 12 * c + (by + 1) * 3 + bx + 1;
        }
        // The window holds 128 words beyond the one it last slid at, and it slides at macroblock boundaries -- enough for the 3200 bits A.3.1 allows
        // a macroblock_layer().  Encoders that ignore the limit exist (very low QP on noisy content: 8000 bits and more), so it also slides here,
        // block by block (a block is at most 25 words): one compare for conforming streams.
        slide_window(e);
        const int nC = kind == 2 ? -1 : nc_of(static_cast<uint8_t>(RDL(vnn, la)), static_cast<uint8_t>(RDL(vnn, lb)));
        const int n = cavlc_residual(e, dst, kind, nC, pmul, padd);
        // ---- bookkeeping per block kind ----
        if (step == 0)
            dc |= n ? 1 : 0;
        else if (step <= 16) {
            vnn = nl == own ? static_cast<uint32_t>(n) : vnn;
            if (n) nzmask |= t8x8 ? (0x33 << ((by & 2) * 4 + (bx & 2))) : (1 << (by * 4 + bx));
        } else if (step <= 18)
            dc |= n ? 2 << (step - 17) : 0;
        else
            vnn = nl == own ? static_cast<uint32_t>(n) : vnn;
    }
    if (dc) s->cur_cbf_dc |= static_cast<uint8_t>(dc);
    s->rec.nzmask = static_cast<uint16_t>(nzmask);
    // the counts of this macroblock's blocks back into the grids (what the edge entries and the next macroblock's caches are made from)
    if (nl < 30) {
        const int gx = nl % 6 - 1, gy = nl / 6 - 1;
        if (gx >= 0 && gx < 4 && gy >= 0) s->nnz_c[nl] = static_cast<uint8_t>(vnn);
    } else if (nl >= 32 && nl < 56) {
        const int g = (nl - 32) % 12, gx = g % 3 - 1, gy = g / 3 - 1;
        if (g < 9 && gx >= 0 && gy >= 0) (&s->nnzc_c[0][0])[nl - 32] = static_cast<uint8_t>(vnn);
    }
}

FI int median3(int a, int b, int c) {
    int mn = a < b ? a : b, mx = a < b ? b : a;
    return c < mn ? mn : (c > mx ? mx : c);
}
// shape: 0 median, 1/2 = 16x8 upper/lower, 3/4 = 8x16 left/right
FI void predict_mv(const Ent &e, const int L, int bx, int by, int w, int ref, int shape, int &px, int &py) {
    const Shared *s = e.s;
    int ia = GI(bx - 1, by), ib = GI(bx, by - 1), ic = GI(bx + w, by - 1);
    int ra = s->ref_c[L][ia], rb = s->ref_c[L][ib], rc = s->ref_c[L][ic];
    if (rc == -2) {
        ic = GI(bx - 1, by - 1);
        rc = s->ref_c[L][ic];
    }
    int ax = s->mv_c[L][ia][0], ay = s->mv_c[L][ia][1], bxv = s->mv_c[L][ib][0], byv = s->mv_c[L][ib][1], cx = s->mv_c[L][ic][0], cy = s->mv_c[L][ic][1];
    if (shape == 1 && rb == ref) {
        px = bxv, py = byv;
        return;
    }
    if ((shape == 2 || shape == 3) && ra == ref) {
        px = ax, py = ay;
        return;
    }
    if (shape == 4 && rc == ref) {
        px = cx, py = cy;
        return;
    }
    if (rb == -2 && rc == -2 && ra != -2) rb = rc = ra, bxv = cx = ax, byv = cy = ay;
    int na = ra == ref, nb = rb == ref, nc = rc == ref;
    if (na + nb + nc == 1) {
        px = na ? ax : (nb ? bxv : cx);
        py = na ? ay : (nb ? byv : cy);
        return;
    }
    px = median3(ax, bxv, cx);
    py = median3(ay, byv, cy);
}
FI void set_part(Ent &e, const int L, int bx, int by, int w, int h, int ref, int mvx, int mvy, int dx, int dy) {
    Shared *s = e.s;
    const uint8_t ax = static_cast<uint8_t>(min(abs(dx), 255)), ay = static_cast<uint8_t>(min(abs(dy), 255));
    const int l = LANE, x = l & 3, y = (l >> 2) & 3; // one 4x4 block per lane (lanes 0..15)
    if (l < 16 && x >= bx && x < bx + w && y >= by && y < by + h) {
        const int g = GI(x, y);
        s->ref_c[L][g] = static_cast<int8_t>(ref);
        *reinterpret_cast<uint32_t *>(s->mv_c[L][g]) = (static_cast<uint32_t>(mvx) & 0xffffu) | (static_cast<uint32_t>(mvy) << 16);
        *reinterpret_cast<uint16_t *>(s->mvd_c[L][g]) = static_cast<uint16_t>(ax | (ay << 8));
    }
    LDS_SYNC();
}
#define PART(bx, by, w, h, shape) static_cast<uint16_t>((bx) | ((by) << 2) | (((w)-1) << 4) | (((h)-1) << 6) | ((shape) << 8))

// What a lane does in fill_caches() never changes: worked out once per slice (build_roles) into Shared::role, so that the macroblock loop
// reads one word instead of redoing the divisions and comparisons (about 90 instructions per macroblock).
//   lanes 0..29 (6-wide luma grid): [8:0] byte offset of the neighbour's entry in nb[] (NO_SRC: none) | k << 9 (index in its edge arrays) |
//                                   edge << 11 (its nnz is an edge value) | interior << 12 | topright << 13 (needs mbx + 1 < wmb)
//   lanes 32..49 (two 3x3 chroma grids): [8:0] byte offset of the source nnz byte in nb[] (NO_SRC: none) | interior << 12 | from_left << 14 |
//                                   index in nnzc_c (flat) << 16
//   lanes 50..55 (DC flags): [8:0] byte offset of the neighbour's cbf_dc byte | bit << 9 | from_left << 14
#define NO_SRC 0x1FFu
FI uint32_t build_role(int l) {
    const uint32_t TB = static_cast<uint32_t>(sizeof(TopInfo));
    if (l < 30) {
        const int gx = l % 6 - 1, gy = l / 6 - 1; // block coordinates relative to the macroblock
        uint32_t off = NO_SRC, k = 0, edge = 0, tr = 0;
        if (gy < 0 && gx >= 0 && gx < 4)
            off = NB_TOP * TB, k = static_cast<uint32_t>(gx), edge = 1;
        else if (gx < 0 && gy >= 0)
            off = NB_LEFT * TB, k = static_cast<uint32_t>(gy), edge = 1;
        else if (gy < 0 && gx < 0)
            off = NB_TL * TB, k = 3;
        else if (gy < 0 && gx == 4)
            off = (NB_TOP + 1) * TB, k = 0, tr = 1;
        const uint32_t interior = gx >= 0 && gx < 4 && gy >= 0;
        return off | k << 9 | edge << 11 | interior << 12 | tr << 13;
    }
    if (l >= 32 && l < 50) {
        const int i = l - 32, cpl = i / 9, g = i % 9, gx = g % 3 - 1, gy = g / 3 - 1;
        uint32_t off = NO_SRC, left = 0;
        if (gy < 0 && gx >= 0)
            off = NB_TOP * TB + static_cast<uint32_t>(offsetof(TopInfo, nnz)) + 4 + cpl * 2 + gx;
        else if (gx < 0 && gy >= 0)
            off = NB_LEFT * TB + static_cast<uint32_t>(offsetof(TopInfo, nnz)) + 4 + cpl * 2 + gy, left = 1;
        const uint32_t interior = gx >= 0 && gy >= 0;
        return off | interior << 12 | left << 14 | static_cast<uint32_t>(cpl * 12 + g) << 16;
    }
    if (l >= 50 && l < 56) {
        const uint32_t left = l < 53;
        return ((left ? NB_LEFT : NB_TOP) * TB + static_cast<uint32_t>(offsetof(TopInfo, cbf_dc))) | static_cast<uint32_t>((l - 50) % 3) << 9 | left << 14;
    }
    return NO_SRC;
}

// ------------------------------------------------------------------ per-MB neighbour caches
FI void fill_caches(Ent &e) {
    Shared *s = e.s; // (e.aw / e.bw -- the first dword of the left / upper entry -- were read by the macroblock loop: the skip decision needs only them)
    const bool a_ok = (e.aw & 255) != 0, b_ok = (e.bw & 255) != 0;
    const int cip = e.cip;
    int l = LANE;
    OPAQUE(l); // keeps lane-dependent addresses from being hoisted out of the macroblock loop and spilled
    const uint32_t role = s->role[l], off = role & NO_SRC;
    const uint8_t *nbb = reinterpret_cast<const uint8_t *>(s->nb);
    int coded = 0; // this lane's bit of the CABAC neighbourhood masks: 1 coded, 2 unavailable (see parse_residual_cabac)
    if (l < 30) {
        int8_t ipm = -2;
        uint8_t nnz = 0x80;
        const int k = static_cast<int>((role >> 9) & 3u); // index inside the neighbour's edge arrays
        const bool has = off != NO_SRC && !((role >> 13) & 1u && e.mbx + 1 >= e.wmb);
        const TopInfo *n = reinterpret_cast<const TopInfo *>(nbb + (off != NO_SRC ? off : 0u));
        const bool n_ok = has && n->type != MBT_NONE;
        if (n_ok) {
            const int inter = MB_IS_INTER(n->type);
            if (!(cip && inter)) ipm = (n->type == MBT_I4x4 || n->type == MBT_I8x8) ? n->ipm[k] : static_cast<int8_t>(2);
            if ((role >> 11) & 1u) nnz = n->nnz[k];
        }
#pragma unroll
        for (int L = 0; L < NL; L++) {
            int8_t ref = -2, refi = -2;
            uint8_t mvdx = 0, mvdy = 0;
            int16_t mvx = 0, mvy = 0;
            if (n_ok) {
                if (MB_IS_INTER(n->type)) {
                    ref = refi = n->ref[L][k >> 1];
                    mvx = n->mv[L][k][0], mvy = n->mv[L][k][1];
                    mvdx = n->mvd[L][k][0], mvdy = n->mvd[L][k][1];
                    if (MI_ENT_B && ((n->dmask >> (k >> 1)) & 1)) refi = 0;
                } else
                    ref = refi = -1;
            }
            s->ref_c[L][l] = ref;
            s->refi_c[L][l] = refi;
            s->mv_c[L][l][0] = mvx, s->mv_c[L][l][1] = mvy;
            s->mvd_c[L][l][0] = mvdx, s->mvd_c[L][l][1] = mvdy;
        }
        if ((role >> 12) & 1u) { // interior: current MB, nothing decoded yet
            nnz = 0;
            ipm = -1;
        }
        coded = (nnz & 0x80) ? 2 : (nnz != 0);
        s->ipm_c[l] = ipm;
        e.v_ipm = ipm;
        s->nnz_c[l] = nnz;
    } else if (l >= 32 && l < 50) {
        const bool ok = off != NO_SRC && (((role >> 14) & 1u) ? a_ok : b_ok);
        const uint8_t v = ((role >> 12) & 1u) ? static_cast<uint8_t>(0) : (ok ? nbb[off != NO_SRC ? off : 0u] : static_cast<uint8_t>(0x80));
        coded = (v & 0x80) ? 2 : (v != 0);
        (&s->nnzc_c[0][0])[(role >> 16) & 31u] = v;
    } else if (l >= 50 && l < 56) { // DC coded_block_flags of the left (50..52) / upper (53..55) macroblock: Intra16x16 luma, Cb, Cr
        const bool ok = ((role >> 14) & 1u) ? a_ok : b_ok;
        coded = ok ? (nbb[off] >> ((role >> 9) & 3u)) & 1 : 2;
    } else if (l >= 56) {
        s->refs8[(l - 56) >> 2][l & 3] = -1;
        if (l < 60) s->sub_type[l - 56] = 0;
        if (l == 60) s->cur_cbf_dc = 0;
    }
    s->coded[l] = static_cast<uint8_t>(coded);
    // (the coefficient staging block is all zero here: it is zeroed once per slice, and a macroblock clears the blocks it filled when it has stored them)
    LDS_SYNC();
}

#if MI_ENT_B
// ------------------------------------------------------------------ direct prediction 8.4.1.2 (B_Skip, B_Direct_16x16, B_Direct_8x8)
FI int min_positive(int a, int b) { return (a >= 0 && b >= 0) ? min(a, b) : max(a, b); }
// Tables 7-14 / 7-18 as packed constants: prediction modes (bit 0 Pred_L0, bit 1 Pred_L1) of the two partitions of
// mb_type 4..21, and mode / shape (0 8x8, 1 8x4, 2 4x8, 3 4x4) of the 13 sub_mb_types (mode 0 = direct)
FI int b_pair_modes(int k) { return static_cast<int>((0xFB7ED69A5ull >> (4 * k)) & 15); } // m0 | m1 << 2 of {1,1},{2,2},{1,2},{2,1},{1,3},{2,3},{3,1},{3,2},{3,3}
FI int b_sub_mode(int st) { return static_cast<int>((0x39FA5E4u >> (2 * st)) & 3); }     // {0,1,2,3,1,1,2,2,3,3,1,2,3}
FI int b_sub_shape(int st) { return static_cast<int>((0x3F99900u >> (2 * st)) & 3); }    // {0,0,0,0,1,2,1,2,1,2,3,3,3}
static_assert(((0x39FA5E4u >> 8) & 3) == 1 && ((0x39FA5E4u >> 24) & 3) == 3 && ((0x39FA5E4u >> 12) & 3) == 2 && ((0x3F99900u >> 8) & 3) == 1 && ((0x3F99900u >> 10) & 3) == 2,
              "Table 7-18 constants");

// Motion of the 8x8 quadrants in `mask8`, one 4x4 block per lane.  commit: straight into the neighbour caches (the whole
// macroblock is direct); otherwise into the staging arrays, from where commit_direct() takes each quadrant when its turn in
// the partition order comes -- until then it must stay "not yet decoded" for the quadrants before it (6.4.11.7).
FI void direct_pred(Ent &e, int mask8, bool commit) {
    Shared *s = e.s;
    int l = LANE;
    OPAQUE(l);
    if (l < 20) s->col[l] = e.v_col;
    LDS_SYNC();
    const int bx = l & 3, by = (l >> 2) & 3, q = (by >> 1) * 2 + (bx >> 1);
    const int cb = e.d8inf ? (by >> 1) * 12 + (bx >> 1) * 3 : (l & 15); // the corner block stands for the quadrant
    const uint32_t mvw = s->col[cb];
    const int mcx = static_cast<int16_t>(mvw & 0xffffu), mcy = static_cast<int32_t>(mvw) >> 16;
    const int refcol = static_cast<int8_t>((s->col[18] >> (8 * q)) & 255u);
    const int slotcol = static_cast<int16_t>((s->col[16 + (q >> 1)] >> (16 * (q & 1))) & 0xffffu);
    int ref0, ref1, m0x, m0y, m1x, m1y;
    if (e.direct_spatial) { // 8.4.1.2.2: reference indices and predictors are those of the MACROBLOCK (neighbours A, B, C outside it)
        int rf[2], px[2] = {0, 0}, py[2] = {0, 0};
#pragma unroll
        for (int L = 0; L < 2; L++) {
            const int ra = s->ref_c[L][GI(-1, 0)], rb = s->ref_c[L][GI(0, -1)];
            int rc = s->ref_c[L][GI(4, -1)];
            if (rc == -2) rc = s->ref_c[L][GI(-1, -1)];
            rf[L] = RFL(min_positive(ra, min_positive(rb, rc)));
        }
        if (rf[0] < 0 && rf[1] < 0)
            rf[0] = rf[1] = 0; // directZeroPredictionFlag
        else {
            if (rf[0] >= 0) predict_mv(e, 0, 0, 0, 4, rf[0], 0, px[0], py[0]);
            if (rf[1] >= 0) predict_mv(e, 1, 0, 0, 4, rf[1], 0, px[1], py[1]);
        }
        const bool colzero = e.col_short && refcol == 0 && mcx >= -1 && mcx <= 1 && mcy >= -1 && mcy <= 1;
        ref0 = rf[0], ref1 = rf[1];
        const bool z0 = rf[0] < 0 || (rf[0] == 0 && colzero), z1 = rf[1] < 0 || (rf[1] == 0 && colzero);
        m0x = z0 ? 0 : px[0], m0y = z0 ? 0 : py[0];
        m1x = z1 ? 0 : px[1], m1y = z1 ? 0 : py[1];
    } else { // 8.4.1.2.3: the picture the co-located block refers to, as the lowest index of this slice's RefPicList0
        ref0 = 0;
        if (refcol >= 0) {
            ref0 = -1;
            for (int i = e.nref - 1; i >= 0; i--)
                if (s->ref_slot[0][i] == slotcol) ref0 = i;
        }
        if (UNI(l < 16 && ((mask8 >> q) & 1) && ref0 < 0)) e.err = 41; // the co-located reference is not in RefPicList0
        ref0 = max(ref0, 0);
        const int dsf = s->dsf[ref0];
        m0x = (dsf * mcx + 128) >> 8, m0y = (dsf * mcy + 128) >> 8;
        m1x = m0x - mcx, m1y = m0y - mcy;
        ref1 = 0;
    }
    if (l < 16 && ((mask8 >> q) & 1)) {
        const uint32_t w0 = (static_cast<uint32_t>(m0x) & 0xffffu) | (static_cast<uint32_t>(m0y) << 16);
        const uint32_t w1 = (static_cast<uint32_t>(m1x) & 0xffffu) | (static_cast<uint32_t>(m1y) << 16);
        if (commit) {
            const int g = GI(bx, by);
            s->ref_c[0][g] = static_cast<int8_t>(ref0), s->ref_c[1][g] = static_cast<int8_t>(ref1);
            *reinterpret_cast<uint32_t *>(s->mv_c[0][g]) = w0, *reinterpret_cast<uint32_t *>(s->mv_c[1][g]) = w1;
        } else {
            *reinterpret_cast<uint32_t *>(s->dmv[0][l]) = w0, *reinterpret_cast<uint32_t *>(s->dmv[1][l]) = w1;
        }
        if (!((bx | by) & 1)) {
            if (commit)
                s->refs8[0][q] = static_cast<int8_t>(ref0), s->refs8[1][q] = static_cast<int8_t>(ref1);
            else
                s->dref[0][q] = static_cast<int8_t>(ref0), s->dref[1][q] = static_cast<int8_t>(ref1);
        }
    }
    LDS_SYNC();
}
// list L of the direct-predicted quadrant q becomes visible to the partitions after it
FI void commit_direct(Ent &e, const int L, int q) {
    Shared *s = e.s;
    const int l = LANE, bx = l & 3, by = (l >> 2) & 3;
    if (l < 16 && (by >> 1) * 2 + (bx >> 1) == q) {
        const int g = GI(bx, by);
        s->ref_c[L][g] = s->dref[L][q];
        *reinterpret_cast<uint32_t *>(s->mv_c[L][g]) = *reinterpret_cast<const uint32_t *>(s->dmv[L][l]);
        if (!((bx | by) & 1)) s->refs8[L][q] = s->dref[L][q];
    }
    LDS_SYNC();
}
#endif

// ref_idx_lX of a partition (7.3.5.1 / 7.3.5.2; te(v) or 9.3.3.1.1.6), entered into the context cache and the per-8x8 list
FI void read_ref_idx(Ent &e, const int L, int bx, int by, int w, int h, int nref) {
    Shared *s = e.s;
    int ref = 0;
    if (nref > 1) {
        if (e.cabac) {
            int ctx = (s->refi_c[L][GI(bx - 1, by)] > 0) + 2 * (s->refi_c[L][GI(bx, by - 1)] > 0);
            while (BIN_A(e, 54 + ctx)) {
                ctx = (ctx >> 2) + 4;
                if (++ref > 31) {
                    e.err = 3;
                    break;
                }
            }
        } else
            ref = nref == 2 ? !get_bit(e) : static_cast<int>(get_ue(e));
        if (ref >= nref || ref >= MI_MAX_REFS) e.err = 13, ref = 0;
    }
    { // one 4x4 block per lane: ref_idx cache for the ctxIdxInc of later partitions, and the per-8x8 list
        const int l = LANE, x = l & 3, y = (l >> 2) & 3;
        if (l < 16 && x >= bx && x < bx + w && y >= by && y < by + h) {
            s->refi_c[L][GI(x, y)] = static_cast<int8_t>(ref);
            if (!((x | y) & 1)) s->refs8[L][(y >> 1) * 2 + (x >> 1)] = static_cast<int8_t>(ref);
        }
        LDS_SYNC();
    }
}
// mvd_lX of a partition (se(v) or UEG3 9.3.2.3 / 9.3.3.1.1.7), prediction 8.4.1.3, result into the caches
FI void read_mv(Ent &e, const int L, int p) {
    Shared *s = e.s;
    const int bx = p & 3, by = (p >> 2) & 3, w = ((p >> 4) & 3) + 1, h = ((p >> 6) & 3) + 1, shape = (p >> 8) & 7;
    const int ref = s->refs8[L][(by >> 1) * 2 + (bx >> 1)];
    int d[2];
    for (int comp = 0; comp < 2; comp++) {
        int v;
        if (e.cabac) { // UEG3, uCoff 9, signed (9.3.2.3, 9.3.3.1.1.7)
            const int sum = s->mvd_c[L][GI(bx - 1, by)][comp] + s->mvd_c[L][GI(bx, by - 1)][comp];
            const int base = comp ? 47 : 40;
            v = 0;
            if (BIN_A(e, base + (sum > 2) + (sum > 32))) {
                int ctx = base + 3;
                v = 1;
                while (v < 9 && BIN_A(e, ctx)) {
                    if (v < 4) ctx++;
                    v++;
                }
                if (v >= 9) v += cabac_egk(e, 3);
                if (cabac_bypass(e)) v = -v;
            }
        } else
            v = get_se(e);
        d[comp] = v;
    }
    int px, py;
    predict_mv(e, L, bx, by, w, ref, shape, px, py);
    set_part(e, L, bx, by, w, h, ref, px + d[0], py + d[1], d[0], d[1]);
}

// ------------------------------------------------------------------ P_Skip without the general machinery
// A skipped macroblock of a P slice has no syntax: its vector comes from the neighbours A, B, C / D (8.4.1.1, 8.4.1.3 with refIdx 0), its
// record and its edge entry are that vector sixteen / four times over next to constants.  Everything is taken straight from the four
// neighbour entries (no neighbour caches are built, no record is staged in LDS): lanes 0..11 hold one dword of the new edge entry, lanes
// 32..63 one dword of the record.  A quarter of the macroblocks of a typical P picture take this path.
#if !MI_ENT_B
FI void pskip_fast(Ent &e) {
    Shared *s = e.s;
    int l = LANE;
    OPAQUE(l);
    const uint32_t *nbw = reinterpret_cast<const uint32_t *>(s->nb); // TOP_DW dwords per entry: [0] type.., [5] ref_idx, [6..9] vectors
    const uint32_t tA = e.aw, tB = e.bw, tC = nbw[(NB_TOP + 1) * TOP_DW], tD = nbw[NB_TL * TOP_DW];
    const uint32_t rA = nbw[NB_LEFT * TOP_DW + 5], rB = nbw[NB_TOP * TOP_DW + 5], rC = nbw[(NB_TOP + 1) * TOP_DW + 5], rD = nbw[NB_TL * TOP_DW + 5];
    const uint32_t mA = nbw[NB_LEFT * TOP_DW + 6], mB = nbw[NB_TOP * TOP_DW + 6], mC = nbw[(NB_TOP + 1) * TOP_DW + 6], mD = nbw[NB_TL * TOP_DW + 9];
    // what fill_caches() would enter for the four blocks: refIdx (-2 unavailable, -1 intra) and vector (0 unless inter)
    auto ref_of = [](uint32_t t, uint32_t r, int byte) {
        const int type = static_cast<int>(t & 255u);
        return type == MBT_NONE ? -2 : (MB_IS_INTER(type) ? static_cast<int>(static_cast<int8_t>((r >> (8 * byte)) & 255u)) : -1);
    };
    auto mv_of = [](uint32_t t, uint32_t m) { return MB_IS_INTER(static_cast<int>(t & 255u)) ? m : 0u; };
    const int ra = ref_of(tA, rA, 0), rb = ref_of(tB, rB, 0);
    const bool c_ok = e.mbx + 1 < e.wmb && (tC & 255u) != MBT_NONE; // C, else D (6.4.11.7)
    const int rc = c_ok ? ref_of(tC, rC, 0) : ref_of(tD, rD, 1);
    const uint32_t va = mv_of(tA, mA), vb = mv_of(tB, mB), vc = c_ok ? mv_of(tC, mC) : mv_of(tD, mD);
    uint32_t mvw = 0;
    const bool zero = ra == -2 || rb == -2 || (ra == 0 && va == 0) || (rb == 0 && vb == 0); // 8.4.1.1
    if (!zero) { // 8.4.1.3.1 with refIdx 0: the one neighbour that uses picture 0, else the median
        const int na = ra == 0, nb_ = rb == 0, nc = rc == 0;
        if (na + nb_ + nc == 1)
            mvw = na ? va : (nb_ ? vb : vc);
        else {
            const int ax = static_cast<int16_t>(va & 0xffffu), ay = static_cast<int32_t>(va) >> 16, bx = static_cast<int16_t>(vb & 0xffffu), by = static_cast<int32_t>(vb) >> 16;
            const int cx = static_cast<int16_t>(vc & 0xffffu), cy = static_cast<int32_t>(vc) >> 16;
            mvw = (static_cast<uint32_t>(median3(ax, bx, cx)) & 0xffffu) | (static_cast<uint32_t>(median3(ay, by, cy)) << 16);
        }
    }
    e.cur_type = MBT_PSKIP;
    e.prev_dqp_nz = 0;
    // (MbRec::avail -- the intra-prediction neighbour availability -- is 0 in the record of an inter macroblock: only K3 reads it, for intra ones)
    // ---- the new edge entry (the same for the row below and for the macroblock to the right), the window on the row above moves on ----
    if (l < TOP_DW) {
        const uint32_t row = MI_ENT_FMO ? static_cast<uint32_t>(e.mby + 1) << 16 : 0u;
        const uint32_t nw = l == 0 ? static_cast<uint32_t>(MBT_PSKIP) : (l == 1 ? row : (l == 2 ? 0xFFFFFFFFu : ((l >= 6 && l < 10) ? mvw : 0u)));
        uint32_t *tl = reinterpret_cast<uint32_t *>(&s->nb[NB_TL]), *tp = reinterpret_cast<uint32_t *>(&s->nb[NB_TOP]), *tr = reinterpret_cast<uint32_t *>(&s->nb[NB_TOP + 1]);
        const uint32_t old_top = tp[l], w1 = tr[l];
        tl[l] = old_top;
        tp[l] = w1;
        tr[l] = e.pre_top;
        reinterpret_cast<uint32_t *>(&s->nb[NB_LEFT])[l] = nw;
        reinterpret_cast<uint32_t *>(e.top + e.mbx)[l] = nw;
        e.pre_top = top_load(e, e.mbx + 3, l);
    } else if (l >= 32) { // ---- the record: one dword per lane ----
        const int k = l - 32;
        const uint32_t t = s->skip_tmpl[k];
        const uint32_t w = k == 0 ? (static_cast<uint32_t>(MBT_PSKIP) | e.qpw0) : (k == 1 ? e.qpw1 : ((k >= 12 && k < 28) ? mvw : t));
        const uint64_t mbi = e.mb_base + static_cast<uint64_t>(e.mby) * e.wmb + e.mbx;
        reinterpret_cast<uint32_t *>(e.mbrec + mbi)[k] = w;
    }
    LDS_SYNC();
}
#endif

// ------------------------------------------------------------------ macroblock_layer() 7.3.5
FI void decode_mb(Ent &e, int skipped) {
    Shared *s = e.s;
    MbRec &r = s->rec;
    const int cabac = e.cabac, islice = e.islice;
    int cbp_luma = 0, cbp_chroma = 0, t8x8 = 0, i16mode = 0, chroma_mode = 0, has_coef = 0;
    int type, raw = 0, nparts = 0;
    const Nb a{e.aw}, b{e.bw};
    r.nzmask = 0;
#if MI_ENT_B
    int no_sub8 = 1; // NoSubMbPartSizeLessThan8x8Flag (7.3.5)
    e.direct8 = 0;
#endif
    if (skipped) {
#if MI_ENT_B
        type = MBT_BSKIP; // the whole macroblock is predicted in direct mode, no residual (7.3.4, 8.4.1.2)
        e.cur_type = type;
        e.direct8 = 15;
        direct_pred(e, 15, true);
#else
        type = MBT_PSKIP;
        e.cur_type = type;
        int mvx = 0, mvy = 0; // 8.4.1.1
        int ra = s->ref_c[0][GI(-1, 0)], rb = s->ref_c[0][GI(0, -1)];
        bool zero = ra == -2 || rb == -2 || (ra == 0 && s->mv_c[0][GI(-1, 0)][0] == 0 && s->mv_c[0][GI(-1, 0)][1] == 0) ||
                    (rb == 0 && s->mv_c[0][GI(0, -1)][0] == 0 && s->mv_c[0][GI(0, -1)][1] == 0);
        if (!zero) predict_mv(e, 0, 0, 0, 4, 0, 0, mvx, mvy);
        set_part(e, 0, 0, 0, 4, 4, 0, mvx, mvy, 0, 0);
        s->refs8[0][0] = s->refs8[0][1] = s->refs8[0][2] = s->refs8[0][3] = 0;
#endif
        e.prev_dqp_nz = 0;
    } else {
        MI_S(e, 3);
        // ---- mb_type (Tables 9-36 / 9-37) ----
        int intra_prefix = 1; // in P slices: bin 0 of mb_type says "intra"
        if (cabac) {
#if MI_ENT_B
            { // Table 9-37 (b): B slices, ctxIdxOffset 27; the intra types hang off the prefix 111101 with their suffix at 32
                intra_prefix = 0;
                const int inc = (a.ok() && a.type() != MBT_BSKIP && a.type() != MBT_BDIRECT) + (b.ok() && b.type() != MBT_BSKIP && b.type() != MBT_BDIRECT);
                if (!BIN_A(e, 27 + inc))
                    raw = 0; // B_Direct_16x16
                else if (!BIN_A(e, 27 + 3))
                    raw = 1 + BINI_A(e, 27 + 5); // B_L0_16x16, B_L1_16x16
                else {
                    int bits = BINI_A(e, 27 + 4) << 3;
                    bits |= BINI_A(e, 27 + 5) << 2;
                    bits |= BINI_A(e, 27 + 5) << 1;
                    bits |= BINI_A(e, 27 + 5);
                    if (bits < 8)
                        raw = bits + 3; // B_Bi_16x16 .. B_L1_L0_16x8
                    else if (bits == 13)
                        intra_prefix = 1;
                    else if (bits == 14)
                        raw = 11; // B_L1_L0_8x16
                    else if (bits == 15)
                        raw = 22; // B_8x8
                    else
                        raw = ((bits << 1) | BINI_A(e, 27 + 5)) - 4; // B_L0_Bi_16x8 .. B_Bi_Bi_8x16
                }
            }
#else
            if (!islice) {
                intra_prefix = BINI_A(e, 14);
                if (!intra_prefix) raw = BIN_A(e, 15) ? 2 - BINI_A(e, 17) : 3 * BINI_A(e, 16);
            }
#endif
            if (intra_prefix) {
                // I-slice bin string; `base` 3 with neighbour-dependent first bin, or the suffix at 17 (P) / 32 (B)
                int base = islice ? 3 : (MI_ENT_B ? 32 : 17), it = 0, first;
                if (islice) {
                    int inc = (a.ok() && a.type() != MBT_I4x4 && a.type() != MBT_I8x8) + (b.ok() && b.type() != MBT_I4x4 && b.type() != MBT_I8x8);
                    first = BINI_A(e, base + inc);
                    base += 2;
                } else
                    first = BINI_A(e, base);
                if (first) {
                    if (cabac_terminate(e))
                        it = 25;
                    else {
                        it = 1 + 12 * BINI_A(e, base + 1);
                        if (BIN_A(e, base + 2)) it += 4 + 4 * BINI_A(e, base + 2 + islice);
                        it += 2 * BINI_A(e, base + 3 + islice);
                        it += BINI_A(e, base + 3 + 2 * islice);
                    }
                }
                raw = islice ? it : it + (MI_ENT_B ? 23 : 5);
            }
        } else
            raw = static_cast<int>(get_ue(e));
        const int it = islice ? raw : raw - (MI_ENT_B ? 23 : 5);
#if MI_ENT_B
        if (raw < 23)
            type = raw == 0 ? MBT_BDIRECT : MBT_B;
#else
        if (!islice && raw < 5)
            type = raw == 0 ? MBT_P16x16 : (raw == 1 ? MBT_P16x8 : (raw == 2 ? MBT_P8x16 : MBT_P8x8));
#endif
        else if (it == 0)
            type = MBT_I4x4;
        else if (it >= 1 && it <= 24) {
            type = MBT_I16x16;
            i16mode = (it - 1) & 3;
            cbp_chroma = ((it - 1) >> 2) % 3;
            cbp_luma = it >= 13 ? 15 : 0;
            if (cbp_chroma && e.mono) e.err = 25, cbp_chroma = 0; // (no such mb_type in a monochrome stream)
        } else if (it == 25)
            type = MBT_IPCM;
        else {
            e.err = 20;
            type = MBT_I4x4;
        }
        e.cur_type = type;
        MI_S(e, 0);
        if (type == MBT_IPCM) {
            // after the terminate bin the arithmetic decoder has consumed exactly what the encoder's
            // flush wrote (9.3.1.2 / 9.3.4.5): stream position = bits fetched - lookahead
            uint32_t pos = bitpos(e);
            if (cabac) pos -= static_cast<uint32_t>(RFL(e.avail));
            seek(e, (pos + 7) & ~7u);
            uint32_t *pcm = reinterpret_cast<uint32_t *>(s->coef);
            for (int i = 0; i < 96; i++) pcm[i] = (e.mono && i >= 64) ? 0x80808080u : __builtin_bswap32(get_bits(e, 32)); // 384 sample bytes in stream order (monochrome: 256; chroma 128)
            if (cabac) cabac_start(e);
            for (int i = 0; i < 16; i++) {
                s->nnz_c[GI(i & 3, i >> 2)] = 16, s->ref_c[0][GI(i & 3, i >> 2)] = -1;
                if (MI_ENT_B) s->ref_c[NL - 1][GI(i & 3, i >> 2)] = -1;
            }
            for (int i = 0; i < 8; i++) s->nnzc_c[i >> 2][(((i >> 1) & 1) + 1) * 3 + (i & 1) + 1] = 16;
            s->cur_cbf_dc = 7;
            r.nzmask = 0xFFFF;
            cbp_luma = 15, cbp_chroma = 2;
            has_coef = 1;
            e.prev_dqp_nz = 0;
        } else {
            if (MB_IS_INTER(type)) {
#if MI_ENT_B
                if (type == MBT_BDIRECT) {
                    e.direct8 = 15;
                    direct_pred(e, 15, true);
                    no_sub8 = e.d8inf;
                } else {
                    // ---- partition schedule (Tables 7-14, 7-18): geometry | prediction modes << 11 ----
                    int nref_parts;
                    if (raw <= 3) {
                        s->parts[0] = static_cast<uint16_t>(PART(0, 0, 4, 4, 0) | (raw << 11));
                        nparts = nref_parts = 1;
                    } else if (raw < 22) {
                        const int mm = b_pair_modes((raw - 4) >> 1), m0 = mm & 3, m1 = mm >> 2;
                        if (raw & 1)
                            s->parts[0] = static_cast<uint16_t>(PART(0, 0, 2, 4, 3) | (m0 << 11)), s->parts[1] = static_cast<uint16_t>(PART(2, 0, 2, 4, 4) | (m1 << 11));
                        else
                            s->parts[0] = static_cast<uint16_t>(PART(0, 0, 4, 2, 1) | (m0 << 11)), s->parts[1] = static_cast<uint16_t>(PART(0, 2, 4, 2, 2) | (m1 << 11));
                        nparts = nref_parts = 2;
                    } else {
                        nref_parts = 4;
                        for (int i = 0; i < 4; i++) {
                            int st;
                            if (cabac) { // Table 9-38 (b), ctxIdxOffset 36
                                if (!BIN_A(e, 36))
                                    st = 0; // B_Direct_8x8
                                else if (!BIN_A(e, 37))
                                    st = 1 + BINI_A(e, 39);
                                else {
                                    st = 3;
                                    int done = 0;
                                    if (BIN_A(e, 38)) {
                                        if (BIN_A(e, 39))
                                            st = 11 + BINI_A(e, 39), done = 1; // B_L1_4x4, B_Bi_4x4
                                        else
                                            st += 4;
                                    }
                                    if (!done) {
                                        st += 2 * BINI_A(e, 39);
                                        st += BINI_A(e, 39);
                                    }
                                }
                            } else
                                st = static_cast<int>(get_ue(e));
                            if (st > 12) e.err = 21, st = 1;
                            s->sub_type[i] = static_cast<int8_t>(st);
                            const int md = b_sub_mode(st), shp = b_sub_shape(st);
                            const int bx = (i & 1) * 2, by = (i >> 1) * 2;
                            if (md == 0) {
                                e.direct8 |= 1 << i;
                                if (!e.d8inf) no_sub8 = 0;
                                s->parts[nparts++] = PART(bx, by, 2, 2, 0); // modes 0: marks the quadrant's turn in the vector loops
                            } else {
                                if (shp) no_sub8 = 0;
                                const int sw = (shp == 0 || shp == 1) ? 2 : 1, sh = (shp == 0 || shp == 2) ? 2 : 1;
                                for (int yy = 0; yy < 2; yy += sh)
                                    for (int xx = 0; xx < 2; xx += sw) s->parts[nparts++] = static_cast<uint16_t>(PART(bx + xx, by + yy, sw, sh, 0) | (md << 11));
                            }
                        }
                        if (e.direct8) direct_pred(e, e.direct8, false); // derived first, visible to the others in partition order
                    }
                    // ---- every ref_idx_l0, every ref_idx_l1, every mvd_l0, every mvd_l1 (7.3.5.1 / 7.3.5.2) ----
#pragma unroll
                    for (int L = 0; L < 2; L++)
                        for (int i = 0; i < nref_parts; i++) {
                            int bx, by, w, h, md;
                            if (raw == 22)
                                bx = (i & 1) * 2, by = (i >> 1) * 2, w = 2, h = 2, md = b_sub_mode(s->sub_type[i]);
                            else {
                                const int p = s->parts[i];
                                bx = p & 3, by = (p >> 2) & 3, w = ((p >> 4) & 3) + 1, h = ((p >> 6) & 3) + 1, md = p >> 11;
                            }
                            if ((md >> L) & 1) read_ref_idx(e, L, bx, by, w, h, L ? e.nref1 : e.nref);
                        }
#pragma unroll
                    for (int L = 0; L < 2; L++)
                        for (int i = 0; i < nparts; i++) {
                            const int p = s->parts[i], md = p >> 11;
                            const int bx = p & 3, by = (p >> 2) & 3;
                            if (md == 0)
                                commit_direct(e, L, (by >> 1) * 2 + (bx >> 1));
                            else if ((md >> L) & 1)
                                read_mv(e, L, p);
                            else // the partition does not use this list: an available neighbour with refIdxLX = -1 and a zero vector (8.4.1.3.2)
                                set_part(e, L, bx, by, ((p >> 4) & 3) + 1, ((p >> 6) & 3) + 1, -1, 0, 0, 0, 0);
                        }
                }
#else
                // ---- partition schedule (Tables 7-13, 7-17) ----
                int nref_parts;
                if (type == MBT_P16x16) {
                    s->parts[0] = PART(0, 0, 4, 4, 0);
                    nparts = nref_parts = 1;
                } else if (type == MBT_P16x8) {
                    s->parts[0] = PART(0, 0, 4, 2, 1), s->parts[1] = PART(0, 2, 4, 2, 2);
                    nparts = nref_parts = 2;
                } else if (type == MBT_P8x16) {
                    s->parts[0] = PART(0, 0, 2, 4, 3), s->parts[1] = PART(2, 0, 2, 4, 4);
                    nparts = nref_parts = 2;
                } else {
                    nref_parts = 4;
                    for (int i = 0; i < 4; i++) {
                        int st;
                        if (cabac) // Table 9-38
                            st = BIN_A(e, 21) ? 0 : (!BIN_A(e, 22) ? 1 : (BIN_A(e, 23) ? 2 : 3));
                        else
                            st = static_cast<int>(get_ue(e));
                        if (st > 3) e.err = 21, st = 0;
                        s->sub_type[i] = static_cast<int8_t>(st);
                        const int bx = (i & 1) * 2, by = (i >> 1) * 2;
                        const int sw = (st == 0 || st == 1) ? 2 : 1, sh = (st == 0 || st == 2) ? 2 : 1;
                        for (int yy = 0; yy < 2; yy += sh)
                            for (int xx = 0; xx < 2; xx += sw) s->parts[nparts++] = PART(bx + xx, by + yy, sw, sh, 0);
                    }
                }
                // ---- ref_idx_l0 per macroblock partition (7.3.5.1 / 7.3.5.2) ----
                for (int i = 0; i < nref_parts; i++) {
                    int bx, by, w, h;
                    if (type == MBT_P8x8)
                        bx = (i & 1) * 2, by = (i >> 1) * 2, w = 2, h = 2;
                    else {
                        const int p = s->parts[i];
                        bx = p & 3, by = (p >> 2) & 3, w = ((p >> 4) & 3) + 1, h = ((p >> 6) & 3) + 1;
                    }
                    read_ref_idx(e, 0, bx, by, w, h, raw != 4 ? e.nref : 1);
                }
                // ---- mvd_l0 + prediction per (sub-)partition ----
                for (int i = 0; i < nparts; i++) read_mv(e, 0, s->parts[i]);
#endif
            } else {
                // ---- intra: transform_size_8x8_flag, prediction modes, intra_chroma_pred_mode ----
                if (type == MBT_I4x4 && e.t8x8_mode) {
                    t8x8 = cabac ? BINI_T8(e, (a.ok() && a.t8x8()) + (b.ok() && b.t8x8())) : static_cast<int>(get_bit(e));
                    if (t8x8) type = MBT_I8x8, e.cur_type = type;
                }
                if (type == MBT_I4x4 || type == MBT_I8x8) {
                    const int n = type == MBT_I8x8 ? 4 : 16;
                    for (int i = 0; i < n; i++) {
                        int bx, by;
                        if (n == 4)
                            bx = (i & 1) * 2, by = (i >> 1) * 2;
                        else
                            bx = (i & 1) + 2 * ((i >> 2) & 1), by = ((i >> 1) & 1) + 2 * (i >> 3);
                        const int pa = static_cast<int>(RDL(e.v_ipm, GI(bx - 1, by))), pb = static_cast<int>(RDL(e.v_ipm, GI(bx, by - 1)));
                        const int pred = (pa < -1 || pb < -1) ? 2 : (pa < pb ? pa : pb); // 8.3.1.1: dcPredModePredictedFlag
                        int mode = pred;
                        if (cabac) {
                            if (!BIN_B(e, 68)) {
                                int rem = 0;
                                for (int k = 0; k < 3; k++) rem |= BINI_B(e, 69) << k;
                                mode = rem < pred ? rem : rem + 1;
                            }
                        } else if (!get_bit(e)) {
                            int rem = static_cast<int>(get_bits(e, 3));
                            mode = rem < pred ? rem : rem + 1;
                        }
                        const int g = GI(bx, by), d = LANE - g;
                        e.v_ipm = (d == 0 || (n == 4 && (d == 1 || d == 6 || d == 7))) ? mode : e.v_ipm;
                    }
                    if (LANE < 30) s->ipm_c[LANE] = static_cast<int8_t>(e.v_ipm); // for the record / neighbour write-out
                }
                if (e.mono)
                    chroma_mode = 0; // ChromaArrayType 0: no intra_chroma_pred_mode (K3 predicts the DC of planes that are 128 everywhere)
                else if (cabac) {
                    int inc = (a.ok() && MB_IS_INTRA(a.type()) && a.type() != MBT_IPCM && a.chroma_mode() != 0) +
                              (b.ok() && MB_IS_INTRA(b.type()) && b.type() != MBT_IPCM && b.chroma_mode() != 0);
                    chroma_mode = 0;
                    if (BIN_B(e, 64 + inc)) {
                        chroma_mode = 1;
                        while (chroma_mode < 3 && BIN_B(e, 67)) chroma_mode++;
                    }
                } else {
                    chroma_mode = static_cast<int>(get_ue(e));
                    if (chroma_mode > 3) e.err = 22, chroma_mode = 0;
                }
                if (LANE < 16) {
                    s->ref_c[0][GI(LANE & 3, LANE >> 2)] = -1;
                    if (MI_ENT_B) s->ref_c[NL - 1][GI(LANE & 3, LANE >> 2)] = -1;
                }
            }
            MI_S(e, MB_IS_INTER(type) ? 1 : 2);
            // ---- coded_block_pattern ----
            if (type != MBT_I16x16) {
                int cbp;
                if (cabac) { // 9.3.3.1.1.4
                    const int cbp_a = a.ok() ? (a.type() == MBT_IPCM ? 0x2F : a.cbp()) : 0x0F;
                    const int cbp_b = b.ok() ? (b.type() == MBT_IPCM ? 0x2F : b.cbp()) : 0x0F;
                    cbp = 0;
                    for (int b8 = 0; b8 < 4; b8++) {
                        int ca = (b8 & 1) ? (cbp >> (b8 - 1)) & 1 : (cbp_a >> (b8 + 1)) & 1;
                        int cb = (b8 & 2) ? (cbp >> (b8 - 2)) & 1 : (cbp_b >> (b8 + 2)) & 1;
                        cbp |= BINI_B(e, 73 + (!ca) + 2 * (!cb)) << b8;
                    }
                    int ca = a.ok() && (a.type() == MBT_IPCM || (a.cbp() >> 4) != 0), cb = b.ok() && (b.type() == MBT_IPCM || (b.cbp() >> 4) != 0);
                    if (!e.mono && BIN_B(e, 77 + ca + 2 * cb)) { // (ChromaArrayType 0: the prefix only)
                        ca = a.ok() && (a.type() == MBT_IPCM || (a.cbp() >> 4) == 2);
                        cb = b.ok() && (b.type() == MBT_IPCM || (b.cbp() >> 4) == 2);
                        cbp |= (1 + BINI_B(e, 77 + 4 + ca + 2 * cb)) << 4;
                    }
                } else {
                    uint32_t k = get_ue(e);
                    if (k > (e.mono ? 15u : 47u)) e.err = 23, k = 0;
                    if (e.mono) k += 48; // the ChromaArrayType 0 column of Table 9-4 (h264/bit_reader.go:118-135) sits behind the 48 entries of the other
                    cbp = RFL(static_cast<int>(MB_IS_INTRA(type) ? e.tab->me_intra[k] : e.tab->me_inter[k]));
                }
                cbp_luma = cbp & 15, cbp_chroma = cbp >> 4;
                if (cbp_luma && e.t8x8_mode && MB_IS_INTER(type)) {
#if MI_ENT_B
                    const int all8 = no_sub8;
#else
                    int all8 = 1;
                    if (type == MBT_P8x8)
                        for (int i = 0; i < 4; i++) all8 &= s->sub_type[i] == 0;
#endif
                    if (all8) t8x8 = cabac ? BINI_T8(e, (a.ok() && a.t8x8()) + (b.ok() && b.t8x8())) : static_cast<int>(get_bit(e));
                }
            }
            // ---- mb_qp_delta + residual ----
            if (cbp_luma || cbp_chroma || type == MBT_I16x16) {
                int dqp;
                if (cabac) { // 9.3.2.7 / 9.3.3.1.1.5
                    int ctx = e.prev_dqp_nz ? 1 : 0, val = 0;
                    while (BIN_A(e, 60 + ctx)) {
                        ctx = 2 + (ctx >> 1);
                        if (++val > 104) {
                            e.err = 2;
                            break;
                        }
                    }
                    dqp = (val & 1) ? (val + 1) >> 1 : -((val + 1) >> 1);
                } else
                    dqp = get_se(e);
                if (dqp < -26 || dqp > 25) e.err = 24, dqp = 0;
                e.prev_dqp_nz = dqp != 0;
                if (dqp) set_qp(e, (e.qp + dqp + 52) % 52);
                MI_S(e, 2);
                MI_T(e, 1);
                MI_R0(e);
                if (cabac)
                    parse_residual_cabac(e, cbp_luma, cbp_chroma, t8x8);
                else
                    parse_residual_cavlc(e, cbp_luma, cbp_chroma, t8x8);
                MI_R(e, 3);
                MI_T(e, 2);
                has_coef = 1;
            } else
                e.prev_dqp_nz = 0;
        }
    }
    MI_T(e, 1);
    // ---- scalar fields of the record ----
    const int qp_store = type == MBT_IPCM ? 0 : e.qp;
    r.type = static_cast<uint8_t>(type);
    r.t8x8 = static_cast<uint8_t>(t8x8);
    r.qp = static_cast<uint8_t>(qp_store);
    r.qpc[0] = static_cast<uint8_t>(RDL(e.v_qpc, min(max(qp_store + e.cqp_off0, 0), 51)));
    r.qpc[1] = static_cast<uint8_t>(RDL(e.v_qpc, min(max(qp_store + e.cqp_off1, 0), 51)));
    r.cbp = static_cast<uint8_t>(cbp_luma | (cbp_chroma << 4));
    r.chroma_mode = static_cast<uint8_t>(chroma_mode);
    r.i16mode = static_cast<uint8_t>(i16mode);
    {
        // neighbour availability for intra prediction (6.4.x; constrained_intra_pred 8.3.1.2)
        const int cip = e.cip;
        int av = 0;
        if (MB_IS_INTRA(type)) { // (only K3 reads it, for intra macroblocks: an inter macroblock's record says 0)
            const int td = s->nb[NB_TL].type, tc = nb_ok(e, NB_TOP + 1) ? s->nb[NB_TOP + 1].type : MBT_NONE;
            if (a.ok() && !(cip && MB_IS_INTER(a.type()))) av |= MI_AV_LEFT;
            if (b.ok() && !(cip && MB_IS_INTER(b.type()))) av |= MI_AV_TOP;
            if (td != MBT_NONE && !(cip && MB_IS_INTER(td))) av |= MI_AV_TOPLEFT;
            if (tc != MBT_NONE && !(cip && MB_IS_INTER(tc))) av |= MI_AV_TOPRIGHT;
        }
        r.avail = static_cast<uint8_t>(av);
    }
    // dbf_idc / alpha_off / beta_off / slice_in_pic / slice_idx of the record are slice constants, written once at slice start
    LDS_SYNC();
    // ---- parallel part: per-block arrays of the record, write-out, neighbour state update ----
    int l = LANE;
    OPAQUE(l); // keeps lane-dependent addresses from being hoisted out of the macroblock loop and spilled
    const int inter = MB_IS_INTER(type);
    // ---- coefficient blocks: which of the 26 staging blocks carry anything, and where they go in the pool ----
    uint32_t cmask = 0, coff = 0;
    if (has_coef) {
        bool nz = false;
        if (l < MI_COEF_BLOCKS) {
            const uint4 a = reinterpret_cast<const uint4 *>(s->coef)[2 * l], b = reinterpret_cast<const uint4 *>(s->coef)[2 * l + 1];
            nz = (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) != 0;
        }
        cmask = static_cast<uint32_t>(__builtin_amdgcn_ballot_w64(nz));
        if (type == MBT_IPCM) cmask = 0xFFFu; // all 384 sample bytes, zero or not
        const uint32_t n = static_cast<uint32_t>(__builtin_popcount(cmask));
        if (n && e.coef_cur + n > e.coef_end) { // next chunk of the pool (a macroblock never straddles chunks)
            uint32_t base = 0;
            if (l == 0) base = atomicAdd(e.pool_head, static_cast<uint32_t>(MI_COEF_CHUNK));
            base = RFL(base);
            if (base + MI_COEF_CHUNK > e.pool_blocks) {
                e.err = 40; // coefficient pool exhausted (H264MI_EDECODE; raise the pool size)
                cmask = 0;
            } else
                e.coef_cur = base, e.coef_end = base + MI_COEF_CHUNK;
        }
        coff = e.coef_cur;
        e.coef_cur += static_cast<uint32_t>(__builtin_popcount(cmask));
    }
    if (l == 0) r.coef_off = coff, r.coef_mask = cmask;
    if (l < 16) {
        int g = GI(l & 3, l >> 2);
        // inter macroblocks have no intra modes: [0..3] = ref_idx_l1 (B; MBREC_REF1), [4] = mb_type as coded (Tables 7-13 / 7-14; 0 when
        // skipped), [5..8] = sub_mb_type of the four 8x8 quadrants (P_8x8 / B_8x8) -- what h264/slice.go:77-102 SliceData keeps
        int8_t v = s->ipm_c[g];
        if (inter) v = (MI_ENT_B && l < 4) ? s->refs8[NL - 1][l] : (l == 4 ? static_cast<int8_t>(raw) : ((l >= 5 && l < 9) ? s->sub_type[l - 5] : static_cast<int8_t>(0)));
        r.ipm[l] = v;
        r.mv[l][0] = inter ? s->mv_c[0][g][0] : static_cast<int16_t>(0);
        r.mv[l][1] = inter ? s->mv_c[0][g][1] : static_cast<int16_t>(0);
    } else if (l < 20) {
        int i = l - 16, ref = inter ? s->refs8[0][i] : -1;
        r.ref[i] = static_cast<int8_t>(ref);
        r.refslot[i] = ref >= 0 ? s->ref_slot[0][ref & (MI_MAX_REFS - 1)] : static_cast<int16_t>(-1);
    }
#if MI_ENT_B
    else if (l < 24) {
        const int i = l - 20, ref = inter ? s->refs8[1][i] : -1;
        r.refslot1[i] = ref >= 0 ? s->ref_slot[1][ref & (MI_MAX_REFS - 1)] : static_cast<int16_t>(-1);
    }
#endif
    // remember the row-above entry of this column for the next MB's top-left neighbour, then build the new one
    TopInfo *tp = &s->nb[NB_TOP];
    if (l >= 32 && l < 32 + TOP_DW) reinterpret_cast<uint32_t *>(&s->nb[NB_TL])[l - 32] = reinterpret_cast<const uint32_t *>(tp)[l - 32];
    LDS_SYNC();
    if (l < 2) {
        TopInfo *dst = &s->nb[l == 0 ? NB_TOP : NB_LEFT];
        dst->type = static_cast<uint8_t>(type);
        dst->t8x8 = static_cast<uint8_t>(t8x8);
        dst->cbp = r.cbp;
        dst->chroma_mode = static_cast<uint8_t>(chroma_mode);
        dst->cbf_dc = s->cur_cbf_dc;
        if (MI_ENT_FMO) dst->row = static_cast<uint16_t>(e.mby + 1);
#pragma unroll
        for (int L = 0; L < NL; L++) {
            dst->ref[L][0] = inter ? s->refs8[L][l == 0 ? 2 : 1] : static_cast<int8_t>(-1);
            dst->ref[L][1] = inter ? s->refs8[L][3] : static_cast<int8_t>(-1);
        }
#if MI_ENT_B
        // direct-predicted 8x8 blocks on the edge: bottom row = quadrants 2, 3; right column = quadrants 1, 3
        dst->dmask = static_cast<uint8_t>(l == 0 ? (e.direct8 >> 2) & 3 : ((e.direct8 >> 1) & 1) | ((e.direct8 >> 2) & 2));
#endif
    } else if (l >= 8 && l < 16) {
        // edge arrays: lanes 8..11 -> top (bottom row), 12..15 -> left (right column)
        int k = l & 3, is_left = l >= 12;
        TopInfo *dst = &s->nb[is_left ? NB_LEFT : NB_TOP];
        int g = is_left ? GI(3, k) : GI(k, 3);
        dst->ipm[k] = s->ipm_c[g];
        dst->nnz[k] = s->nnz_c[g];
#pragma unroll
        for (int L = 0; L < NL; L++) {
            dst->mv[L][k][0] = s->mv_c[L][g][0], dst->mv[L][k][1] = s->mv_c[L][g][1];
            dst->mvd[L][k][0] = s->mvd_c[L][g][0], dst->mvd[L][k][1] = s->mvd_c[L][g][1];
        }
    } else if (l >= 16 && l < 24) {
        // chroma nnz edges: [plane][k]
        int i = l - 16, is_left = i >= 4, cpl = (i >> 1) & 1, k = i & 1;
        TopInfo *dst = &s->nb[is_left ? NB_LEFT : NB_TOP];
        dst->nnz[4 + cpl * 2 + k] = is_left ? s->nnzc_c[cpl][(k + 1) * 3 + 2] : s->nnzc_c[cpl][2 * 3 + k + 1];
    }
    LDS_SYNC();
    // new entry -> HBM row; slide the LDS window: [0] <- [1], [1] <- prefetched column x+2; prefetch x+3
    if (l < TOP_DW) {
        const uint32_t nw = reinterpret_cast<const uint32_t *>(tp)[l], w1 = reinterpret_cast<const uint32_t *>(&s->nb[NB_TOP + 1])[l];
        reinterpret_cast<uint32_t *>(e.top + e.mbx)[l] = nw;
        reinterpret_cast<uint32_t *>(&s->nb[NB_TOP])[l] = w1;
        reinterpret_cast<uint32_t *>(&s->nb[NB_TOP + 1])[l] = e.pre_top;
        e.pre_top = top_load(e, e.mbx + 3, l);
    }
    const uint64_t mbi = e.mb_base + static_cast<uint64_t>(e.mby) * e.wmb + e.mbx;
    if (l >= 32) // MbRec: 128 bytes = 32 dwords, lanes 32..63
        reinterpret_cast<uint32_t *>(e.mbrec + mbi)[l - 32] = reinterpret_cast<const uint32_t *>(&r)[l - 32];
#if MI_ENT_B
    if (l < 16) reinterpret_cast<uint32_t *>(e.mbmv1 + mbi)[l] = inter ? *reinterpret_cast<const uint32_t *>(s->mv_c[1][GI(l & 3, l >> 2)]) : 0u; // MbMv1
#endif
    if (l < MI_COEF_BLOCKS && ((cmask >> l) & 1)) { // present blocks, packed in ascending order: 2 x 16 bytes per lane
        uint4 *dst = reinterpret_cast<uint4 *>(e.coefs) + 2 * (static_cast<size_t>(coff) + __builtin_popcount(cmask & ((1u << l) - 1u)));
        dst[0] = reinterpret_cast<const uint4 *>(s->coef)[2 * l], dst[1] = reinterpret_cast<const uint4 *>(s->coef)[2 * l + 1];
        // ... and the staging block is all zero again for the next macroblock (only blocks with something in them were ever written)
        reinterpret_cast<uint4 *>(s->coef)[2 * l] = make_uint4(0, 0, 0, 0), reinterpret_cast<uint4 *>(s->coef)[2 * l + 1] = make_uint4(0, 0, 0, 0);
    }
    LDS_SYNC();
}

// All-zero records (type MBT_NONE) for macroblocks [from, to) of the picture: what this slice is responsible for but did
// not decode.  The reconstruction kernels skip them (K3 paints them mid-grey), K5 finds no edge to filter.
FI void fill_none(const Ent &e, int from, int to) {
    const int l = LANE;
    for (int a = from + (l >> 5); a < to; a += 2) reinterpret_cast<uint32_t *>(e.mbrec + e.mb_base + static_cast<uint64_t>(a))[l & 31] = 0u;
}

// ------------------------------------------------------------------ kernel: slice_data() 7.3.4
// grid = number of slices of the launch; `slice_base` = index of its first slice (the host orders the slices by launch)
#if MI_ENT_B
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) k_entropy_b(const SliceDesc *slices_, const PicDesc *pics, const uint8_t *bitstream, const DevTables *tab, MbRec *mbrec,
                                                           int16_t *coefs, uint32_t *pool_head, uint32_t pool_blocks, uint32_t *status_, uint32_t *toprows_, int wmb_max, uint32_t slice_base,
                                                           const BSliceExt *bexts, MbMv1 *mbmv1) {
#else
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MI_ENT_MINWAVES, 8))) MI_ENT_KERNEL(const SliceDesc *slices_, const PicDesc *pics, const uint8_t *bitstream, const DevTables *tab, MbRec *mbrec,
                                                           int16_t *coefs, uint32_t *pool_head, uint32_t pool_blocks, uint32_t *status_, uint32_t *toprows_, int wmb_max, uint32_t slice_base) {
#endif
    __shared__ Shared sh;
    const uint32_t slice_no = slice_base + blockIdx.x;
    const SliceDesc *slices = slices_ + slice_base;
    uint32_t *status = status_ + 8 * static_cast<size_t>(slice_base);
    uint32_t *toprows = toprows_ + static_cast<size_t>(slice_base) * wmb_max * TOP_DW;
    const uint64_t t_begin = wall_clock64();
    Ent e;
    e.lane = static_cast<int>(threadIdx.x);
    e.s = &sh;
#if MI_ENT_STATS
    e.bins = 0;
    for (int k = 0; k < 4; k++) e.tacc[k] = 0;
#endif
    e.top = reinterpret_cast<TopInfo *>(toprows + static_cast<size_t>(blockIdx.x) * wmb_max * TOP_DW);
    e.pre_top = 0;
    e.tab = tab;
    const SliceDesc *sd = &slices[blockIdx.x];
    const PicDesc *pd = &pics[sd->pic_idx];
    e.sd = sd, e.pd = pd;
    e.rbsp32 = reinterpret_cast<const uint32_t *>(bitstream + sd->rbsp_off); // slices are 16-byte aligned in the staging buffer
    e.rbsp_words = RFL((sd->rbsp_size + 3) >> 2);
    e.mbrec = mbrec;
    e.coefs = coefs;
    e.pool_head = pool_head, e.pool_blocks = pool_blocks;
    e.coef_cur = e.coef_end = 0;
    e.err = 0;
    e.cabac = RFL(static_cast<int>(pd->cabac));
    e.islice = RFL(static_cast<int>(sd->slice_type == 2));
#if MI_ENT_ISLICE_PRIO
    // the I slice of a GOP is the longest wavefront of the launch by far: let it win the instruction arbitration
    // against the P-slice wavefronts that share its SIMD
    if (e.islice) __builtin_amdgcn_s_setprio(MI_ENT_ISLICE_PRIO);
#endif
    e.wmb = RFL(static_cast<int>(pd->wmb)), e.hmb = RFL(static_cast<int>(pd->hmb));
    e.cip = RFL(static_cast<int>(pd->cip)), e.t8x8_mode = RFL(static_cast<int>(pd->t8x8_mode));
    e.cqp_off0 = RFL(static_cast<int>(pd->cqp_off[0])), e.cqp_off1 = RFL(static_cast<int>(pd->cqp_off[1]));
    e.nref = RFL(static_cast<int>(sd->num_ref_idx_active));
    e.mb_base = (static_cast<uint64_t>(RFL(static_cast<uint32_t>(pd->mb_base >> 32))) << 32) | RFL(static_cast<uint32_t>(pd->mb_base));
    e.prev_dqp_nz = 0;
    e.range = 510, e.value = 0, e.avail = 0;
    e.mbx = e.mby = 0, e.cur_type = 0;
    const int l = LANE;
    // ---- per-lane tables ----
    { // Tables 9-44 / 9-45: lane p keeps the entries of pStateIdx p (cabac_decide says how the transition word is laid out)
        const uint8_t *rl = tab->range_lps[l];
        e.v_rlps = rl[0] | (rl[1] << 8) | (rl[2] << 16) | (static_cast<uint32_t>(rl[3]) << 24);
        e.v_trans = (static_cast<uint32_t>(tab->trans_lps[l]) ^ static_cast<uint32_t>(l)) | (l == 0 ? 0x80000040u : 0u) | (l < 62 ? 256u : 0u);
    }
    // coefficient scans of the picture (8.5.6, 8.5.7): zig-zag, or the field scan in a field picture (h264/slice.go:867-872 field_pic_flag)
    const uint8_t *scan4 = pd->field ? tab->fieldscan4 : tab->zigzag4, *scan8 = pd->field ? tab->fieldscan8 : tab->zigzag8;
    e.v_maps = (pd->field ? tab->sig8x8_field[l] : tab->sig8x8[l]) | (tab->last8x8[l] << 8) | (scan8[l] << 16) | (static_cast<uint32_t>(scan4[l & 15]) << 24);
    e.v_pos = scan4[l & 15] | (scan4[(l + 1) & 15] << 8) | (scan8[l] << 16) | (static_cast<uint32_t>(l) << 24);
    e.mono = RFL(static_cast<int>(pd->mono));
    e.v_cat0 = cat_word0(l), e.v_cat1 = cat_word1(l, pd->field != 0); // (CAVLC slices: v_cat0 is replaced by the run_before tables below, one entry per lane)
    e.v_qpc = tab->qpc[l < 52 ? l : 51];
    set_qp(e, RFL(static_cast<int>(sd->slice_qp)));
    e.v_step = step_word(l);
    e.aw = e.bw = 0;
    e.v_ipm = 0;
    sh.posmap[0][l] = scan4[l & 15];
    sh.posmap[1][l] = scan4[(l + 1) & 15];
    sh.posmap[2][l] = scan8[l];
    sh.posmap[3][l] = static_cast<uint8_t>(l);
    if (l < MI_MAX_REFS) sh.ref_slot[0][l] = sd->ref_slot[l];
#if MI_ENT_B
    {
        const BSliceExt *bx = &bexts[sd->bext];
        if (l < MI_MAX_REFS) sh.ref_slot[1][l] = bx->ref_slot1[l], sh.dsf[l] = bx->dist_scale[l];
        e.nref1 = RFL(static_cast<int>(bx->num_ref_idx_l1_active));
        e.direct_spatial = RFL(static_cast<int>(bx->direct_spatial)), e.d8inf = RFL(static_cast<int>(bx->direct_8x8_inference));
        e.col_short = RFL(static_cast<int>(bx->col_short));
        e.col = reinterpret_cast<const uint32_t *>(bx->col);
        e.direct8 = 0, e.v_col = 0;
        e.mbmv1 = mbmv1;
    }
#endif
    { // context variables 9.3.1.1: macroblock-level states into the two VGPRs, residual states into LDS
        const int set = e.islice ? 0 : 1 + sd->cabac_init_idc;
        const uint8_t *src = tab->ctx_init[set][sd->slice_qp];
        e.ca = ctx_word(src[l]);
        e.cb = ctx_word(src[l < 61 ? 64 + l : 399 + (l - 61)]);
        e.wk = 0;
        e.wk_cat = -1, e.wk_c0 = 0, e.wk_home = 0, e.wk_valid = 0;
        for (int i = l; i < 464; i += 64) sh.ctx[i] = src[i];
    }
    if (!pd->cabac) // CAVLC: the code tables move into LDS (2.2 KB; a lookup in HBM-resident direct tables was most of a CAVLC slice's time)
        for (int i = l; i < MI_VLC_N / 2; i += 64) reinterpret_cast<uint32_t *>(sh.vlc)[i] = reinterpret_cast<const uint32_t *>(tab->vlc_c)[i];
    if (l < 32) reinterpret_cast<uint32_t *>(&sh.rec)[l] = 0;
    LDS_SYNC();
    if (!pd->cabac) e.v_cat0 = l < 48 ? sh.vlc[MI_VLC_RUN + l] : 0u; // run_before for zerosLeft 1..6: 6 x 8 entries, read with v_readlane
    if (l == 0) { // slice constants of every MbRec
        sh.rec.dbf_idc = sd->dbf_idc;
        sh.rec.alpha_off = sd->alpha_off, sh.rec.beta_off = sd->beta_off;
        sh.rec.slice_in_pic = sd->slice_in_pic;
        sh.rec.slice_idx = slice_no;
        sh.rec.refslot1[0] = sh.rec.refslot1[1] = sh.rec.refslot1[2] = sh.rec.refslot1[3] = -1; // (the B build rewrites them per macroblock)
    }
    if (l < 32) { // the record of a P_Skip macroblock, as far as it is a constant of the slice (pskip_fast adds QP, availability and the vector)
        const uint32_t slot0 = static_cast<uint16_t>(sd->ref_slot[0]);
        sh.skip_tmpl[l] = l == 2 ? static_cast<uint32_t>(sd->dbf_idc) << 24
                                 : (l == 3 ? (static_cast<uint32_t>(static_cast<uint8_t>(sd->alpha_off)) | static_cast<uint32_t>(static_cast<uint8_t>(sd->beta_off)) << 8 |
                                              static_cast<uint32_t>(sd->slice_in_pic) << 16)
                                           : ((l == 9 || l == 10) ? (slot0 | slot0 << 16) : (l == 11 ? slice_no : (l >= 30 ? 0xFFFFFFFFu : 0u))));
    }
    for (int i = l; i < MI_COEF_PER_MB / 2; i += 64) reinterpret_cast<uint32_t *>(sh.coef)[i] = 0; // the coefficient staging block: zero between macroblocks
    sh.role[l] = build_role(l);
    for (int i = l; i < e.wmb * TOP_DW; i += 64) reinterpret_cast<uint32_t *>(e.top)[i] = 0; // all row-above entries: type NONE
    if (l < TOP_DW) reinterpret_cast<uint32_t *>(&sh.nb[NB_LEFT])[l] = 0, reinterpret_cast<uint32_t *>(&sh.nb[NB_TL])[l] = 0;
    if (l < 2 * TOP_DW) reinterpret_cast<uint32_t *>(&sh.nb[NB_TOP])[l] = 0;
    LDS_SYNC();
    {
        uint32_t pos = RFL(sd->data_bit_off);
        if (e.cabac) pos = (pos + 7) & ~7u; // cabac_alignment_one_bit
        e.wbase = 0x80000000u; // force the window load
        seek(e, pos);
        if (e.cabac) cabac_start(e);
    }
    const int total = min(e.wmb * e.hmb, RFL(static_cast<int>(sd->end_mb))); // the next slice's territory is out of bounds
    const uint32_t stop_bit = RFL(sd->stop_bit);
    int addr = RFL(static_cast<int>(sd->first_mb));
    // Slice groups (FMO, 8.2.2; h264/slice.go:134-158, :530-552): the picture's mbToSliceGroupMap follows the slices in the
    // bitstream buffer; the slice walks the macroblocks of its group (nextMbAddress), the host has zeroed all records.
    // (only two flags live across the macroblock loop: what the slow path needs beyond them is fetched again where it is used)
    const bool fmo = MI_ENT_FMO && RFL(static_cast<int>(pd->fmo)) != 0;
    bool consecutive = false; // FMO: the previous macroblock of the slice is (mbx - 1, mby)
    if (!fmo) fill_none(e, RFL(static_cast<int>(sd->fill_from)), min(addr, total)); // a gap in front of the first slice of the picture
    e.mbx = addr % e.wmb, e.mby = addr / e.wmb;
    int more = 1, skip_state = 0 /* 0: read mb_skip_run, 1: inside a run, 2: coded MB follows a run */, pending = 0;
    int n_mbs = 0;
    while (more && !e.err) {
        if (addr >= total) {
            e.err = 30;
            break;
        }
        if (fmo) {
            // The previous macroblock of the slice is the left neighbour only if it is (mbx - 1, mby); the row-above window
            // cannot slide (the next column is anywhere), so the three entries above are fetched for every macroblock, and an
            // entry is a neighbour only if this slice wrote it in the row above (TopInfo::row).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wavefront's stores to e.top[] have reached L2
            if (!consecutive && l < TOP_DW) {
                reinterpret_cast<uint32_t *>(&sh.nb[NB_LEFT])[l] = 0;
                reinterpret_cast<uint32_t *>(&sh.nb[NB_TL])[l] = e.mbx > 0 ? top_load(e, e.mbx - 1, l) : 0u;
            }
            if (l < 2 * TOP_DW) reinterpret_cast<uint32_t *>(&sh.nb[NB_TOP])[l] = top_load(e, e.mbx + l / TOP_DW, l % TOP_DW);
            LDS_SYNC();
            if (l < 3 && sh.nb[NB_TL + l].row != static_cast<uint16_t>(e.mby)) sh.nb[NB_TL + l].type = MBT_NONE;
            LDS_SYNC();
        } else if (e.mbx == 0 || n_mbs == 0) { // new MB row (or slice start): no left / top-left neighbour; (re)load the row-above window
            if (l < TOP_DW) reinterpret_cast<uint32_t *>(&sh.nb[NB_LEFT])[l] = 0, reinterpret_cast<uint32_t *>(&sh.nb[NB_TL])[l] = 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // stores of the previous row to e.top[] have been issued to L2
            if (l < 2 * TOP_DW) reinterpret_cast<uint32_t *>(&sh.nb[NB_TOP])[l] = top_load(e, e.mbx + l / TOP_DW, l % TOP_DW);
            if (l < TOP_DW) e.pre_top = top_load(e, e.mbx + 2, l);
            LDS_SYNC();
        }
        OPAQUE(e.lane);
        slide_window(e);
        MI_T0(e);
#if MI_ENT_B
        // the co-located macroblock's record (80 bytes), consumed by direct_pred(); "intra" when there is no such picture
        e.v_col = (e.col && l < 20) ? e.col[static_cast<size_t>(addr) * 20 + l] : ((l >= 16 && l < 19) ? 0xFFFFFFFFu : 0u);
#endif
        e.aw = RFL(*reinterpret_cast<const uint32_t *>(&sh.nb[NB_LEFT])), e.bw = RFL(*reinterpret_cast<const uint32_t *>(&sh.nb[NB_TOP]));
        int skipped = 0;
        if (!e.islice) {
            if (e.cabac) {
                const Nb a{e.aw}, b{e.bw};
                const int skip_type = MI_ENT_B ? MBT_BSKIP : MBT_PSKIP; // ctxIdxOffset 11 in P slices, 24 in B slices (Table 9-34)
                skipped = BINI_A(e, (MI_ENT_B ? 24 : 11) + (a.ok() && a.type() != skip_type) + (b.ok() && b.type() != skip_type));
            } else {
                if (skip_state == 0) {
                    pending = static_cast<int>(get_ue(e));
                    if (pending > total - addr) e.err = 31, pending = 0;
                    if (pending > 0) skip_state = 1;
                }
                if (skip_state == 1) {
                    skipped = 1;
                    pending--;
                }
            }
        }
#if !MI_ENT_B
        if (skipped) {
            MI_T(e, 1);
            pskip_fast(e);
        } else
#endif
        {
            MI_T(e, 1);
            fill_caches(e);
            MI_T(e, 0);
            decode_mb(e, skipped);
        }
        MI_T(e, 3);
        if (e.err) break; // the record of this macroblock cannot be trusted: it is blanked with the rest of the range
        n_mbs++;
        if (e.cabac)
            more = !cabac_terminate(e);
        else if (skipped) {
            if (pending == 0) {
                more = bitpos(e) < stop_bit;
                skip_state = 2;
            }
        } else {
            more = bitpos(e) < stop_bit;
            skip_state = 0;
        }
        if (fmo) { // nextMbAddress (8-17): 64 candidates at a time
            const uint8_t *sgmap = bitstream + RFL(e.pd->sgmap_off);
            const int sgroup = RFL(static_cast<int>(sgmap[RFL(static_cast<int>(e.sd->first_mb))]));
            const uint32_t inv_wmb = RFL(e.pd->inv_wmb);
            int next = total;
            for (int base = addr + 1; base < total; base += 64) {
                const int i = base + l;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(i < total && sgmap[i] == sgroup);
                if (m) {
                    next = base + __builtin_ctzll(m);
                    break;
                }
            }
            consecutive = next == addr + 1 && e.mbx + 1 < e.wmb;
            addr = next;
            e.mby = static_cast<int>(__umulhi(static_cast<uint32_t>(addr), inv_wmb)), e.mbx = addr - e.mby * e.wmb;
        } else {
            addr++;
            if (++e.mbx == e.wmb) e.mbx = 0, e.mby++;
        }
    }
    if (!fmo) fill_none(e, addr, total); // after an error or an early end of the slice
    if (l == 0) {
        status[8 * blockIdx.x] = static_cast<uint32_t>(e.err);
        status[8 * blockIdx.x + 1] = static_cast<uint32_t>(n_mbs);
        status[8 * blockIdx.x + 2] = static_cast<uint32_t>(wall_clock64() - t_begin);
        status[8 * blockIdx.x + 3] = MI_BINS(e);
#if MI_ENT_STATS
        for (int k = 0; k < 4; k++) status[8 * blockIdx.x + 4 + k] = static_cast<uint32_t>(e.tacc[k] >> 4); // shader clocks / 16
#if MI_ENT_STATS == 3 /* when the wavefront ran (100 MHz ticks, low 32 bits): the launch's timeline by slice type (H264MI_SLICE_TIMELINE) */
        status[8 * blockIdx.x + 4] = static_cast<uint32_t>(t_begin), status[8 * blockIdx.x + 5] = static_cast<uint32_t>(wall_clock64());
#endif
#endif
    }
}

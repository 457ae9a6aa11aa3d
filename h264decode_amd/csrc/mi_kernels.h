// h264decode_amd/csrc/mi_kernels.h -- kernel entry points (HIP, gfx950) and launch helpers.
#pragma once
#include <hip/hip_runtime.h>
#include "mi_types.h"

// K1/K2: one slice per wavefront.  grid = #slices, block = 64; toprows = 12 dwords per MB column per slice.
extern "C" __global__ void k_entropy(const SliceDesc *slices, const PicDesc *pics, const uint8_t *bitstream, const DevTables *tab, MbRec *mbrec, int16_t *coefs,
                                     uint32_t *pool_head, uint32_t pool_blocks, uint32_t *status, uint32_t *toprows, int wmb_max, uint32_t slice_base);
// the same with the slice-group walk of 8.2.2 (k_entropy_f.hip): for launches that hold a picture with more than one slice group
extern "C" __global__ void k_entropy_f(const SliceDesc *slices, const PicDesc *pics, const uint8_t *bitstream, const DevTables *tab, MbRec *mbrec, int16_t *coefs,
                                       uint32_t *pool_head, uint32_t pool_blocks, uint32_t *status, uint32_t *toprows, int wmb_max, uint32_t slice_base);
// the same for B slices (k_entropy_b.hip): two reference lists, direct prediction from the ColRec array of RefPicList1[0]; toprows = 18 dwords per column
extern "C" __global__ void k_entropy_b(const SliceDesc *slices, const PicDesc *pics, const uint8_t *bitstream, const DevTables *tab, MbRec *mbrec, int16_t *coefs,
                                       uint32_t *pool_head, uint32_t pool_blocks, uint32_t *status, uint32_t *toprows, int wmb_max, uint32_t slice_base,
                                       const BSliceExt *bexts, MbMv1 *mbmv1);
#define MI_TOPROW_BYTES 72 /* per macroblock column per slice (the B kernel's TopInfo; the I/P kernel uses 48 of them) */
#ifndef MI_K4_WG
#define MI_K4_WG 1 /* wavefronts (groups of four macroblocks) per workgroup of k_inter / k_inter_b: grid = ceil(n_blocks / MI_K4_WG) rounded up to a multiple of 8 (measured: 2 and 4 are slower, profiles/r05_k4_budget.txt) */
#endif
// K4 (k_inter.hip): inter macroblocks of a set of pictures (one per stream), one lane per 4x4 block, four macroblocks per wavefront.
// n_blocks = #pictures << groups_per_pic_log2 (groups of four macroblocks per picture, rounded up to a power of two: no division in the kernel);
// grid = n_blocks rounded up to a multiple of 8 (XCD-aware block order), block = 64
extern "C" __global__ void k_inter(const uint32_t *pic_list, const PicDesc *pics, const SliceDesc *slices, const DevTables *tab, const MbRec *mbrec, const int16_t *coefs,
                                   int groups_per_pic_log2, int n_blocks);
// K4 for pictures with B slices: two lists per block (MbRec::refslot1, MbMv1), default / explicit / implicit weighting
extern "C" __global__ void k_inter_b(const uint32_t *pic_list, const PicDesc *pics, const SliceDesc *slices, const DevTables *tab, const MbRec *mbrec,
                                     const int16_t *coefs, int groups_per_pic_log2, int n_blocks, const BSliceExt *bexts, const MbMv1 *mbmv1);
// K3: intra macroblocks, one workgroup per picture, one wavefront per macroblock row (2-D wavefront order).
extern "C" __global__ void k_intra(const uint32_t *pic_list, const PicDesc *pics, const FramePool *pools, const DevTables *tab, const MbRec *mbrec,
                                   const int16_t *coefs, const unsigned long long *intramask);
// K3 spread over `nbands` workgroups per picture: grid = pictures * nbands, block = 64 * wavefronts per band (<= MI_INTRA_WAVES);
// xdone: pictures * nbands * wmb_max flag words; epoch / ticket as for k_deblock_x
extern "C" __global__ void k_intra_x(const uint32_t *pic_list, const PicDesc *pics, const FramePool *pools, const DevTables *tab, const MbRec *mbrec,
                                     const int16_t *coefs, uint32_t *xdone, uint32_t epoch, int nbands, uint32_t *ticket, uint32_t ticket_base, int wmb_max,
                                     uint32_t *xstatus, int waves_per_row, const unsigned long long *intramask);
// k_dbprep: boundary strengths + alpha / beta / tC0 of every macroblock of a batch (DbPrm), so that K5 -- one serial dependency chain per
// picture -- has none of that work in its steps; for the pictures flagged PicDesc::save_col also their ColRec array (the motion later B pictures
// take their direct prediction from).  grid = (ceil(mbs_max / MI_DBPREP_MBS), pictures of the list), block = 256.
#ifndef MI_DBPREP_MBS
#define MI_DBPREP_MBS 64
#endif
extern "C" __global__ void k_dbprep(const uint32_t *pic_list, const PicDesc *pics, const DevTables *tab, const MbRec *mbrec, const MbMv1 *mbmv1, DbPrm *out, int col_only, unsigned long long *intramask, int n_pics);
// K5 (k_deblock.hip): in-loop deblocking, one workgroup per picture, one wavefront per group of 8 macroblock rows, 8 lanes per macroblock
// (two lines per lane, packed 16-bit arithmetic).  block = 64 * nwaves, dynamic LDS = mi_deblock8_lds_bytes(nwaves, ring, ring_last, last_bufs);
// (nwaves, ring, ring_last, last_bufs) from mi_deblock8_plan()
// xstatus: a wavefront that gives up waiting for its neighbour (4 s) leaves a code there; a -DMI_DB_STATS build adds the phase clocks of its step loop to xstatus[8..19]
extern "C" __global__ void k_deblock(const uint32_t *pic_list, const PicDesc *pics, const DbPrm *dbprm, int ring, int ring_last, int last_bufs, uint32_t *xstatus);
#define MI_DEBLOCK8_MAX_WAVES 8      /* 512 threads: two wavefronts on a SIMD, up to 256 VGPRs each (the kernel holds 84 of them as landing registers of its loads) */
#define MI_DEBLOCK8_MAX_GROUPS 64    /* hmb <= 320 + slack */
#define MI_DEBLOCK8_HDR_BYTES 512    /* sizeof(Db8Shared) */
#define MI_DEBLOCK8_TILE_BYTES 1568  /* the LDS window of a sub-row: four macroblock columns of luma (1024) and chroma (512) + 32 (bank stagger) */
#define MI_DEBLOCK8_WAVE_BYTES (9 * MI_DEBLOCK8_TILE_BYTES + 2 * 8 * 80) /* 8 sub-rows + rows 12..15 of the row above the first; the DbPrm stage (two steps x 8 macroblocks) */
static inline size_t mi_deblock8_lds_bytes(int nwaves, int ring, int ring_last, int last_bufs) {
    return MI_DEBLOCK8_HDR_BYTES + static_cast<size_t>(nwaves) * MI_DEBLOCK8_WAVE_BYTES + (static_cast<size_t>(nwaves - 1) * ring + static_cast<size_t>(ring_last) * last_bufs) * 96;
}
// K5 spread over `nbands` workgroups per picture (k_deblock_x.hip): grid = pictures * nbands, block = 64 * (largest band's group count),
// dynamic LDS = mi_deblock_lds_bytes_banded().  xring: pictures * nbands * wmb_max * 24 granules of 8 bytes; epoch: a value no earlier
// launch on this ring has used (never 0); ticket / ticket_base: a counter that only ever grows and its value before this launch.
extern "C" __global__ void k_deblock_x(const uint32_t *pic_list, const PicDesc *pics, const DbPrm *dbprm, int ring, int ring_last, int last_bufs,
                                       unsigned long long *xring, uint32_t epoch, int nbands, uint32_t *ticket, uint32_t ticket_base, int wmb_max, uint32_t *xstatus,
                                       int roles);
#ifndef MI_DEBLOCK_MAX_WAVES
#define MI_DEBLOCK_MAX_WAVES 12    /* the banded kernel: LDS: 6 KB of row state per wavefront + its hand-off ring */
#endif
#define MI_DEBLOCK_HDR_BYTES 1552  /* sizeof(DbShared) rounded up to 16 */
#define MI_DEBLOCK_WAVE_BYTES 4800 /* 4 sub-rows: sample tiles, the macroblock's DbPrm, the bottom rows for the sub-row below */
#define MI_DEBLOCK_SLOT_BYTES 96
#define MI_DEBLOCK_LDS_MAX (160 * 1024)
// banded builds: one ring region per wavefront (the last one stages what goes to the global ring) + the staging slot of the band above
static inline size_t mi_deblock_lds_bytes_banded(int nwaves, int ring, int wave_bytes = MI_DEBLOCK_WAVE_BYTES) {
    return MI_DEBLOCK_HDR_BYTES + static_cast<size_t>(nwaves) * wave_bytes + (static_cast<size_t>(nwaves) * ring + 1) * MI_DEBLOCK_SLOT_BYTES;
}
// How a launch of n_pics pictures of up to wmb x hmb macroblocks is spread over the chip.  nbands = 1: the one-workgroup kernels.
// Otherwise every band is one round of at most MI_DEBLOCK_MAX_WAVES groups, and pictures * bands stays within `max_wgs`
// workgroups (all of them can be resident at once; the ticket order makes the hand-off safe even if they are not).
static inline void mi_deblock_bands(int n_pics, int wmb, int hmb, int max_wgs, int *nbands, int *nwaves, int *ring, int *roles) {
    const int ngroups = (hmb + 3) / 4;
    int nb = n_pics > 0 ? max_wgs / n_pics : 1;
    if (nb > ngroups) nb = ngroups;
    if (nb < 2 || (ngroups + nb - 1) / nb > MI_DEBLOCK_MAX_WAVES) nb = 1;
    const int per_band = (ngroups + nb - 1) / nb;
    // two wavefronts per group (luma | chroma) while every wavefront of the launch can still have a SIMD to itself (1024 of them)
    *roles = (nb > 1 && 2 * per_band <= MI_DEBLOCK_MAX_WAVES && 2 * n_pics * ngroups <= 1280) ? 2 : 1;
    *nbands = nb;
    *nwaves = per_band * *roles;
    *ring = wmb < 16 ? (wmb > 0 ? wmb : 1) : 16;
}
// Wavefront count and hand-off ring depths of k_deblock for pictures of wmb x hmb macroblocks.  Wavefront w runs the 8-row groups
// w, w + nwaves, ...; group g hands rows 12..15 of its last macroblock row to group g + 1 through the ring region of its wavefront.  A group may
// run at most `depth` columns ahead of the group below it.  Groups of one round run side by side, 8 steps apart, so a
// short ring is enough between them; but the reader of the LAST wavefront's groups is wavefront 0 in the NEXT round, which
// only starts when it has finished a whole row -- that ring (ring_last) holds a whole row, otherwise the last wavefront
// (and through back-pressure every wavefront above it) would wait for wavefront 0.  From three rounds on that ring is
// double-buffered by round: the last wavefront's group of round r must not wait until wavefront 0 has read ALL of round
// r - 1's ring, because wavefront 0's group of round r can only finish when the groups below it -- up to the last
// wavefront's, through the short rings -- make progress: with a single buffer wide pictures deadlock.  As many wavefronts as fit.
// 1080p: 9 wavefronts, one round.
static inline void mi_deblock8_plan(int wmb, int hmb, int *nwaves, int *ring, int *ring_last, int *last_bufs, int max_waves = MI_DEBLOCK8_MAX_WAVES) {
    const int ngroups = (hmb + 7) / 8;
    const int w1 = wmb > 0 ? wmb : 1;
    for (int nw = ngroups < max_waves ? ngroups : max_waves; nw >= 1; nw--) {
        const int rounds = (ngroups + nw - 1) / nw;
        const int r = w1 < 16 ? w1 : 16;
        const int rl = rounds > 1 ? w1 : r;
        const int nb = rounds > 2 ? 2 : 1;
        if (mi_deblock8_lds_bytes(nw, r, rl, nb) <= MI_DEBLOCK_LDS_MAX) {
            *nwaves = nw, *ring = r, *ring_last = rl, *last_bufs = nb;
            return;
        }
    }
    *nwaves = 1, *ring = 1, *ring_last = w1, *last_bufs = 2; // 2 x 512 columns x 96 bytes + one wavefront always fit
}
// K6: crop + tight pack of a list of frames into I420; grid = (frames, ceil(2 * h_max / rows_per_block)), block = 256
typedef struct {
    uint64_t src;     // Y plane of the frame (coded size W x H; Cb at + W*H, Cr at + W*H*5/4)
    uint64_t dst_off; // byte offset of the packed frame in the destination buffer
    uint32_t W, H, x0, y0, w, h;
} PackDesc;
extern "C" __global__ void k_pack(const PackDesc *descs, uint8_t *dst, int rows_per_block);

#ifndef MI_INTRA_WAVES
#define MI_INTRA_WAVES 16 /* 1024 threads, 128 VGPRs per wavefront: the kernels need 120 once nothing lane-dependent is hoisted out of the macroblock loop */
#endif

// K3 over several workgroups per picture: bands of about 4 macroblock rows, as many as keep pictures * bands within max_wgs;
// nbands = 1: the one-workgroup kernel.  p_only (no intra-only picture in the launch): bands of 3 rows with up to 4 wavefronts
// per row -- the few intra macroblocks of a P / B picture are dealt round-robin to them (k_intra_x, waves_per_row).
static inline void mi_intra_bands(int n_pics, int hmb, int max_wgs, bool p_only, int *nbands, int *nwaves, int *wpr) {
    int nb = n_pics > 0 ? max_wgs / n_pics : 1;
    const int want = (hmb + (p_only ? 2 : 3)) / (p_only ? 3 : 4);
    if (nb > want) nb = want;
    if (nb < 2) nb = 1;
    const int rows = (hmb + nb - 1) / nb, slots = rows < MI_INTRA_WAVES ? rows : MI_INTRA_WAVES;
    int w = (nb > 1 && p_only) ? MI_INTRA_WAVES / slots : 1;
    if (w > 4) w = 4;
    if (w < 1) w = 1;
    *nbands = nb;
    *wpr = w;
    *nwaves = nb > 1 ? slots * w : MI_INTRA_WAVES;
}

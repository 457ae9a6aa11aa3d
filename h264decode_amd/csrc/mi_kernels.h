// h264decode_amd/csrc/mi_kernels.h -- kernel entry points (HIP, gfx950) and launch helpers.
#pragma once
#include <hip/hip_runtime.h>
#include "mi_types.h"

// K1/K2: one slice per wavefront.  grid = #slices, block = 64; toprows = 12 dwords per MB column per slice.
extern "C" __global__ void k_entropy(const SliceDesc *slices, const PicDesc *pics, const uint8_t *bitstream, const DevTables *tab, MbRec *mbrec, int16_t *coefs,
                                     uint32_t *status, uint32_t *toprows, int wmb_max);
// K4: inter macroblocks of a set of pictures (one per stream), one macroblock per wavefront.
// n_blocks = #pictures * mbs_per_pic_max; grid = n_blocks rounded up to a multiple of 8 (XCD-aware block order)
extern "C" __global__ void k_inter(const uint32_t *pic_list, const PicDesc *pics, const SliceDesc *slices, const FramePool *pools, const DevTables *tab,
                                   const MbRec *mbrec, const int16_t *coefs, int mbs_per_pic_max, int n_blocks);
// K3: intra macroblocks, one workgroup per picture, one wavefront per macroblock row (2-D wavefront order).
extern "C" __global__ void k_intra(const uint32_t *pic_list, const PicDesc *pics, const FramePool *pools, const DevTables *tab, const MbRec *mbrec,
                                   const int16_t *coefs);
// K5: in-loop deblocking, one workgroup per picture, one wavefront per group of 4 macroblock rows.
// block = 64 * nwaves (2..MI_DEBLOCK_MAX_WAVES, or 1 for a single group), dynamic LDS = mi_deblock_lds_bytes(nwaves)
extern "C" __global__ void k_deblock(const uint32_t *pic_list, const PicDesc *pics, const FramePool *pools, const DevTables *tab, const MbRec *mbrec);
#define MI_DEBLOCK_MAX_WAVES 12   /* 170 VGPRs -> 3 wavefronts per SIMD */
#define MI_DEBLOCK_RING 32        /* macroblock columns of the hand-off ring between row groups (power of two) */
#define MI_DEBLOCK_HDR_BYTES 1088 /* sizeof(DbShared) rounded up to 16 */
#define MI_DEBLOCK_WAVE_BYTES 8832
#define MI_DEBLOCK_SLOT_BYTES 96
static inline size_t mi_deblock_lds_bytes(int nwaves) {
    return MI_DEBLOCK_HDR_BYTES + static_cast<size_t>(nwaves) * (MI_DEBLOCK_WAVE_BYTES + MI_DEBLOCK_RING * MI_DEBLOCK_SLOT_BYTES);
}
// number of wavefronts for a picture of hmb macroblock rows: fewest rounds over the row groups, then fewest wavefronts
static inline int mi_deblock_waves(int hmb) {
    const int ngroups = (hmb + 3) / 4;
    const int rounds = (ngroups + MI_DEBLOCK_MAX_WAVES - 1) / MI_DEBLOCK_MAX_WAVES;
    return (ngroups + rounds - 1) / rounds;
}
// K6: crop + tight pack of one frame into I420
extern "C" __global__ void k_pack(const uint8_t *src_y, const uint8_t *src_cb, const uint8_t *src_cr, int pitch, int x0, int y0, int w, int h, uint8_t *dst);

#define MI_INTRA_WAVES 16

// h264decode_amd/csrc/mi_kernels.h -- kernel entry points (HIP, gfx950) and launch helpers.
#pragma once
#include <hip/hip_runtime.h>
#include "mi_types.h"

// K1/K2: one slice per wavefront.  grid = #slices, block = 64; toprows = 12 dwords per MB column per slice.
extern "C" __global__ void k_entropy(const SliceDesc *slices, const PicDesc *pics, const uint8_t *bitstream, const DevTables *tab, MbRec *mbrec, int16_t *coefs,
                                     uint32_t *status, uint32_t *toprows, int wmb_max);
// K4: inter macroblocks of a set of pictures (one per stream), one macroblock per wavefront.
extern "C" __global__ void k_inter(const uint32_t *pic_list, const PicDesc *pics, const SliceDesc *slices, const FramePool *pools, const DevTables *tab,
                                   const MbRec *mbrec, const int16_t *coefs, int mbs_per_pic_max);
// K3: intra macroblocks, one workgroup per picture, one wavefront per macroblock row (2-D wavefront order).
extern "C" __global__ void k_intra(const uint32_t *pic_list, const PicDesc *pics, const FramePool *pools, const DevTables *tab, const MbRec *mbrec,
                                   const int16_t *coefs);
// K5: in-loop deblocking, one workgroup per picture, one wavefront per macroblock row.
extern "C" __global__ void k_deblock(const uint32_t *pic_list, const PicDesc *pics, const FramePool *pools, const DevTables *tab, const MbRec *mbrec,
                                     int wmb_max);
#define MI_DEBLOCK_SLOT_BYTES 96 /* sizeof(GroupSlot): dynamic LDS = MI_DEBLOCK_WAVES * wmb_max * 96 */
// K6: crop + tight pack of one frame into I420
extern "C" __global__ void k_pack(const uint8_t *src_y, const uint8_t *src_cb, const uint8_t *src_cr, int pitch, int x0, int y0, int w, int h, uint8_t *dst);

#define MI_INTRA_WAVES 16
#define MI_DEBLOCK_WAVES 4 /* each wavefront filters 4 macroblock rows at once */

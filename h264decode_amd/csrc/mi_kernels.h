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
// block = 64 * nwaves, dynamic LDS = mi_deblock_lds_bytes(nwaves, ring); (nwaves, ring) from mi_deblock_plan()
extern "C" __global__ void k_deblock(const uint32_t *pic_list, const PicDesc *pics, const FramePool *pools, const DevTables *tab, const MbRec *mbrec, int ring);
#ifndef MI_DEBLOCK_MAX_WAVES
#define MI_DEBLOCK_MAX_WAVES 9     /* LDS: 14.6 KB of row state per wavefront + its hand-off ring */
#endif
#define MI_DEBLOCK_HDR_BYTES 1088  /* sizeof(DbShared) rounded up to 16 */
#define MI_DEBLOCK_WAVE_BYTES 14976
#define MI_DEBLOCK_SLOT_BYTES 96
#define MI_DEBLOCK_LDS_MAX (160 * 1024)
static inline size_t mi_deblock_lds_bytes(int nwaves, int ring) {
    return MI_DEBLOCK_HDR_BYTES + static_cast<size_t>(nwaves) * (MI_DEBLOCK_WAVE_BYTES + static_cast<size_t>(ring) * MI_DEBLOCK_SLOT_BYTES);
}
// Wavefront count and hand-off ring depth for pictures of wmb x hmb macroblocks.  Fewest rounds over the 4-row groups,
// then fewest wavefronts.  A group may run at most `ring` columns ahead of the group below it, and the group below
// the last wavefront's group only starts when wavefront 0 has finished its first group, so nwaves * ring must cover a
// whole row (otherwise the chain of back-pressure stops wavefront 0 before the end of its row: deadlock).
static inline void mi_deblock_plan(int wmb, int hmb, int *nwaves, int *ring) {
    const int ngroups = (hmb + 3) / 4;
    for (int nw = ngroups < MI_DEBLOCK_MAX_WAVES ? ngroups : MI_DEBLOCK_MAX_WAVES; nw >= 1; nw--) {
        const int rounds = (ngroups + nw - 1) / nw;
        int w = (ngroups + rounds - 1) / rounds; // fewest wavefronts for that many rounds
        if (w < 1) w = 1;
        int r = rounds == 1 ? 16 : (wmb + 16 + w - 1) / w + 8;
        if (r < 16) r = 16;
        if (r > wmb) r = wmb > 0 ? wmb : 1; // a whole row never needs back-pressure
        if (mi_deblock_lds_bytes(w, r) <= MI_DEBLOCK_LDS_MAX) {
            *nwaves = w, *ring = r;
            return;
        }
    }
    *nwaves = 1, *ring = wmb > 0 ? wmb : 1;
}
// K6: crop + tight pack of one frame into I420
extern "C" __global__ void k_pack(const uint8_t *src_y, const uint8_t *src_cb, const uint8_t *src_cr, int pitch, int x0, int y0, int w, int h, uint8_t *dst);

#ifndef MI_INTRA_WAVES
#define MI_INTRA_WAVES 12 /* 768 threads: 170 VGPRs per wavefront (16 wavefronts would cap them at 128 and spill) */
#endif

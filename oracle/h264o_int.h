/* oracle/h264o_int.h -- internal structures of the CPU oracle.  TEST INFRASTRUCTURE ONLY. */
#ifndef H264O_INT_H
#define H264O_INT_H
#include "h264o.h"

/* Inter types name the partition shape; which lists a partition predicts from is in ref[] / ref1[] (>= 0: used), so the
 * shapes serve P and B macroblocks alike.  MBT_BDIRECT = B_Direct_16x16 (with residual), MBT_BSKIP = B_Skip. */
enum { MBT_NONE = 0, MBT_I4x4, MBT_I8x8, MBT_I16x16, MBT_IPCM, MBT_P16x16, MBT_P16x8, MBT_P8x16, MBT_P8x8, MBT_PSKIP, MBT_BDIRECT, MBT_BSKIP };
#define MB_IS_INTRA(t) ((t) >= MBT_I4x4 && (t) <= MBT_IPCM)
#define MB_IS_INTER(t) ((t) >= MBT_P16x16)

typedef struct {
    uint8_t *plane[3];
    int stride[3];
    int frame_num, frame_num_wrap, pic_num, poc;
    int ref; /* 0 unused, 1 short-term, 2 long-term */
    int nonexisting; /* a frame inferred by the gaps-in-frame_num process (8.2.5.2): it fills a place in the window, holds no picture */
    int long_term_frame_idx;
    int id; /* unique, increasing */
    int in_use;
    /* field pictures (PAFF): which fields of the frame are decoded (bit 0 top, bit 1 bottom; a frame picture sets both) and their
     * PicOrderCnt; parity = -1 for a frame, 0 / 1 for the field views (h264o_decoder::fviews) that the lists of field pictures hold */
    int fields, fpoc[2], parity;
    int funref; /* fields marked "unused for reference" on their own (operation 1 in a field picture) while the other one still is one */
    struct h264o_mb_s *mbs; /* motion of the decoded picture (co-located data for direct prediction, 8.4.1.2) */
    int n_mbs;
} h264o_pic;

typedef struct h264o_mb_s {
    uint8_t type; /* MBT_* ; MBT_NONE = not yet decoded in this picture */
    uint8_t t8x8, qp, qpc[2], cbp_luma, cbp_chroma, chroma_mode;
    uint16_t slice_id;
    int8_t ipm[16];   /* Intra4x4/8x8PredMode per 4x4 block, raster (by*4+bx); -1 when not I_NxN */
    uint8_t nnz[24];  /* TotalCoeff per 4x4 block: luma raster [0..15], Cb [16..19], Cr [20..23] */
    uint16_t nzmask;  /* luma blocks (raster) holding non-zero coefficients; 8x8 blocks replicated */
    uint8_t cbf_dc;   /* bit0 Intra16x16 DC, bit1 Cb DC, bit2 Cr DC (CABAC coded_block_flag) */
    int16_t mv[16][2];
    int8_t ref[4];    /* ref_idx per 8x8 (raster 2x2); -1 intra */
    int32_t refid[4]; /* h264o_pic.id referenced, for deblocking */
    int16_t mvd[16][2]; /* |mvd| per 4x4 block for CABAC ctxIdxInc */
    int8_t alpha_off, beta_off; /* FilterOffsetA/B of the containing slice */
    uint8_t dbf_idc;
    uint8_t dqp_nz; /* mb_qp_delta != 0 (CABAC ctxIdxInc of the next MB) */
    /* list 1 (B macroblocks); ref[] / ref1[] = -1 where the list is not used */
    int16_t mv1[16][2];
    int8_t ref1[4];
    int32_t refid1[4];
    int16_t mvd1[16][2];
    uint8_t direct8; /* bit i: 8x8 block i is direct-predicted (B_Skip / B_Direct_16x16 / B_Direct_8x8): ref_idx contexts read it as 0 */
} h264o_mb;

typedef struct {
    int mb_type_raw, type, mbx, mby, addr;
    int i16mode, chroma_mode, cbp_luma, cbp_chroma, t8x8;
    int sub_type[4];
    int16_t i16dc[16];     /* scan order */
    int16_t luma[16][16];  /* [luma4x4BlkIdx (z-order)][scan idx]; Intra16x16 AC uses idx 1..15 */
    int16_t luma8[4][64];  /* [luma8x8BlkIdx][scan idx] */
    int16_t cdc[2][4];
    int16_t cac[2][4][16]; /* [iCbCr][blk][scan idx 1..15] */
    uint8_t pcm[384];
} h264o_curmb;

struct h264o_decoder {
    h264o_sps sps[32];
    h264o_pps pps[256];
    uint8_t *sg_ids[256]; /* slice_group_id[] of the PPSs with slice_group_map_type 6 */
    uint8_t *sgmap;       /* mbToSliceGroupMap of the current picture (NULL: one slice group) */
    int first_mbs[1024], n_first_mbs; /* first_mb_in_slice of the slices of the current picture */
    const h264o_sps *asps;
    const h264o_pps *apps;
    int wmb, hmb;   /* of the current PICTURE: hmb = fhmb for a frame, fhmb / 2 for a field */
    int fhmb;       /* macroblock rows of a frame */
    h264o_pic pics[20];
    /* field views of pics[]: the same samples with the rows of one parity only (plane + parity * stride, stride * 2), their own
     * id and PicOrderCnt -- a field picture is decoded, and predicted from, through these, so that the reconstruction code sees
     * an ordinary picture of half the height */
    h264o_pic fviews[20][2];
    int n_pics;
    h264o_pic *cur;  /* what is being reconstructed: the frame, or the view of the current field */
    h264o_pic *curf; /* the frame it belongs to (== cur for a frame picture): what marking, lists and output deal with */
    h264o_pic *pend; /* a frame whose first field is decoded and whose second may follow */
    int field_pic, bottom, second_field; /* of the current picture */
    const uint8_t *scan4, *scan8; /* coefficient scans of the current picture: zig-zag or field (8.5.6, 8.5.7) */
    int next_pic_id;
    h264o_mb *mb;
    /* POC state (8.2.1) */
    int prev_poc_msb, prev_poc_lsb, prev_frame_num, prev_frame_num_offset, prev_ref_has_mmco5;
    int poc_top, poc_bot; /* TopFieldOrderCnt / BottomFieldOrderCnt of the picture compute_poc() was last asked about */
    int prev_ref_frame_num; /* PrevRefFrameNum (7.4.3): frame_num of the previous reference picture, 0 after an IDR picture or operation 5 */
    /* slice state */
    h264o_slice_header sh;
    h264o_slice_header first_sh; /* first slice header of the current picture */
    h264o_br br;
    int slice_id;
    h264o_pic *rpl0[33];
    h264o_pic *rpl1[33];
    int qp;
    int prev_dqp_nz;
    uint32_t cod_range, cod_offset;
    uint8_t ctx[H264O_NCTX]; /* (pStateIdx << 1) | valMPS */
    int level_scale4[6][6][16]; /* [list][qp%6][raster pos] */
    int level_scale8[2][6][64];
    h264o_curmb c;
    uint16_t cur_done[2]; /* per list: 4x4 blocks (raster) of the current MB whose motion is final */
    int cur_sub;          /* sub-macroblock being decoded (3 when the macroblock is not split): later ones are not available (6.4.11.7) */
    /* output */
    int crop;
    uint8_t *out;
    size_t out_cap, out_pos;
    h264o_stream_info info;
    int32_t *trace;
    size_t trace_cap, trace_pos;
    char err[256];
    uint8_t *rbsp;
    size_t rbsp_cap;
    /* test instrumentation: what the picture-management code actually executed (see h264o_last_features) and the
     * PicOrderCnt of every output picture */
    uint32_t feat;
    int32_t pocs[8192];
    int n_pocs;
};

/* field scans (Table 8-12 / 8-13, field columns) as raster positions (recon.c) */
extern const uint8_t h264o_fieldscan4x4[16], h264o_fieldscan8x8[64];
/* entropy.c */
int h264o_decode_slice_data(h264o_decoder *d);
void h264o_cabac_init_engine(h264o_decoder *d);
int h264o_cabac_decision(h264o_decoder *d, int ctx_idx);
/* recon.c */
void h264o_build_level_scale(h264o_decoder *d);
void h264o_recon_mb(h264o_decoder *d, h264o_mb *m);
void h264o_deblock_picture(h264o_decoder *d);
int h264o_fail(h264o_decoder *d, const char *fmt, ...);

static inline int h264o_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int h264o_clip1(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static inline int h264o_qpc(int qpi) { return qpi < 30 ? qpi : h264o_qpc_tab[qpi - 30]; }
/* luma4x4BlkIdx (z-order) -> raster index by*4+bx (6.4.3) */
static inline int h264o_blk_raster(int idx) { return (((idx >> 1) & 1) + 2 * (idx >> 3)) * 4 + ((idx & 1) + 2 * ((idx >> 2) & 1)); }

#endif

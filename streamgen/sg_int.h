/* streamgen/sg_int.h -- internal declarations of the synthetic stream generator. */
#ifndef SG_INT_H
#define SG_INT_H
#include "sg.h"
#include "sg_tables.h"

/* ---- bit writer / CABAC encoder (sg_bits.c) ---- */
typedef struct {
    uint8_t *buf;
    size_t cap, pos; /* bytes */
    uint32_t acc;
    int nacc; /* bits in acc */
    int overflow;
    /* CABAC encoder state (9.3.4) */
    uint32_t low, range;
    int first_bit, outstanding;
    uint8_t ctx[SG_NCTX]; /* (pStateIdx<<1)|valMPS */
} sg_bw;

void sg_bw_init(sg_bw *w, uint8_t *buf, size_t cap);
void sg_put(sg_bw *w, uint32_t v, int n);
void sg_put_ue(sg_bw *w, uint32_t v);
void sg_put_se(sg_bw *w, int32_t v);
void sg_put_te(sg_bw *w, int range, uint32_t v);
void sg_trailing(sg_bw *w); /* rbsp_trailing_bits */
size_t sg_bw_bytes(sg_bw *w);
int sg_bw_aligned(sg_bw *w);
/* wrap an RBSP into a NAL with start code + emulation prevention; returns bytes written */
size_t sg_write_nal(uint8_t *dst, size_t cap, int long_sc, int ref_idc, int type, const uint8_t *rbsp, size_t n);

void sg_cabac_init_ctx(sg_bw *w, int set, int slice_qp);
void sg_cabac_start(sg_bw *w);
void sg_cabac_bin(sg_bw *w, int ctx, int bin);
void sg_cabac_bypass(sg_bw *w, int bin);
void sg_cabac_terminate(sg_bw *w, int bin); /* bin=1 also flushes */

/* ---- reconstruction primitives (sg_recon.c), written independently of oracle/ ---- */
typedef struct {
    uint8_t *pl[3];
    int w, h; /* coded luma size */
    int id, frame_num;
    int is_ref;   /* 0 unused for reference, 1 short-term, 2 long-term */
    int long_idx; /* LongTermFrameIdx when is_ref == 2 */
    int poc;
    int nonexist; /* a frame that only exists as a skipped frame_num value (8.2.5.2): takes a place in the window, is never predicted from */
    int dropped;     /* field pictures: this field was marked "unused for reference" on its own (memory_management_control_operation 1) */
    int parity, fid; /* field pictures (sg_params::field_pics): 0 top / 1 bottom, and the frame the field belongs to */
    void *motion; /* the picture's macroblock motion (an emb array of sg_enc.c): co-located data of later B pictures */
} sg_pic;

typedef struct {
    int left, top, topleft, topright;
} sg_avail;

int sg_intra_mode_allowed(int kind /*4,8,16,0=chroma*/, int mode, const sg_avail *a);
void sg_pred_i4(const sg_pic *p, int x, int y, int mode, const sg_avail *a, uint8_t *pred /*4x4*/);
void sg_pred_i8(const sg_pic *p, int x, int y, int mode, const sg_avail *a, uint8_t *pred /*8x8*/);
void sg_pred_i16(const sg_pic *p, int x, int y, int mode, const sg_avail *a, uint8_t *pred /*16x16*/);
void sg_pred_chroma(const sg_pic *p, int plane, int x, int y, int mode, const sg_avail *a, uint8_t *pred /*8x8*/);
void sg_mc_luma(const sg_pic *ref, int x, int y, int w, int h, int mvx, int mvy, uint8_t *dst, int dstride);
void sg_mc_chroma(const sg_pic *ref, int plane, int x, int y, int w, int h, int mvx, int mvy, uint8_t *dst, int dstride);

/* levels in scan order -> residual samples (raster).  ls = LevelScale for qP%6, raster order. */
void sg_residual4(const int16_t *lev, const int *ls, int qp, int have_dc, int dc, int *res);
void sg_residual8(const int16_t *lev, const int *ls, int qp, int *res);
void sg_luma_dc(const int16_t *lev_scan, int ls00, int qp, int *dc_raster16);
void sg_chroma_dc(const int16_t *lev4, int ls00, int qpc, int *dc4);
/* least-squares quantisers (projection on the decoder's own basis functions) */
void sg_quant4(const int *resid, const int *ls, int qp, double dead, int skip_dc, int16_t *lev_scan);
void sg_quant8(const int *resid, const int *ls, int qp, double dead, int16_t *lev_scan);
void sg_quant_luma_dc(const int *blk_sums16, int ls00, int qp, double dead, int16_t *lev_scan);
void sg_quant_chroma_dc(const int *blk_sums4, int ls00, int qpc, double dead, int16_t *lev4);

typedef struct {
    uint8_t intra, t8x8, qp, qpc[2], dbf_idc;
    int8_t alpha_off, beta_off;
    uint16_t slice_id, nzmask;
    int16_t mv[16][2];
    int32_t refid[4];
    int16_t mv1[16][2]; /* list 1 of B macroblocks; refid / refid1 = -1 where a list is not used */
    int32_t refid1[4];
} sg_dbmb;
void sg_deblock(sg_pic *p, const sg_dbmb *mbs, int wmb, int hmb);
/* field pictures: coefficients in field scan order (8.5.6, 8.5.7); deblocking with the rules for field macroblocks (8.7.2.1:
 * horizontal edges of intra macroblocks get bS 3, vertical vector differences count from 4 quarter FRAME samples = 2 field ones) */
void sg_set_field_mode(int on);

#endif

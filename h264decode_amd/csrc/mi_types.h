/*
 * h264decode_amd/csrc/mi_types.h -- data layout shared by host code and HIP kernels.
 *
 * HBM layout per decoder (sized once at create time for 288 GB HBM3E parts; everything stays
 * resident, nothing is re-allocated per batch):
 *   bitstream   uint8[]            RBSP bytes of every slice of the batch, 16-byte aligned per slice
 *   slices      SliceDesc[]        one per slice (host-built)            -> entropy kernels
 *   pics        PicDesc[]          one per picture (host-built)          -> all kernels
 *   mbrec       MbRec[]            128 B per macroblock (entropy -> recon/deblock)
 *   coef        packed pool        32-byte blocks (16 int16, raster order inside the block), only the blocks with a non-zero coefficient
 *   dbprm       DbPrm[]            80 B per macroblock (k_dbprep -> K5): boundary strengths, alpha / beta / tC0
 *   frames      per stream: `slots` frames of (coded W x H luma + 2 x W/2 x H/2 chroma), pitch = W
 *   tables      DevTables          CABAC/CAVLC/deblock tables + per-PPS LevelScale sets
 */
#ifndef MI_TYPES_H
#define MI_TYPES_H
#include <stdint.h>

/* B macroblocks: MBT_B stands for every coded inter type of Table 7-14 (the partition geometry only matters while the
 * macroblock is parsed); B_Direct_16x16 and B_Skip are kept apart because neighbours' CABAC contexts ask for them. */
enum { MBT_NONE = 0, MBT_I4x4, MBT_I8x8, MBT_I16x16, MBT_IPCM, MBT_P16x16, MBT_P16x8, MBT_P8x16, MBT_P8x8, MBT_PSKIP, MBT_B, MBT_BDIRECT, MBT_BSKIP };
#define MB_IS_INTRA(t) ((t) >= MBT_I4x4 && (t) <= MBT_IPCM)
#define MB_IS_INTER(t) ((t) >= MBT_P16x16)

#define MI_MAX_REFS 16
/* Coefficient staging layout of a macroblock: 26 blocks of 16 int16 --
 * luma 16 blocks (raster 4x4 blocks, or 4 x 64 for the 8x8 transform) | I16 DC | chroma DC 2x4 + pad 8 | chroma AC 2x4 blocks.
 * The entropy kernels assemble it in LDS; only the blocks with a non-zero coefficient go to HBM, packed back to back in a
 * per-pass pool (MbRec::coef_off / coef_mask), and K3 / K4 scatter them into the same layout in LDS again. */
#define MI_COEF_PER_MB 416
#define MI_COEF_I16DC 256
#define MI_COEF_CDC 272
#define MI_COEF_CAC 288
#define MI_COEF_BLOCKS 26
#define MI_COEF_CHUNK 2048 /* blocks a slice wavefront takes from the pool at a time (64 KB) */

/* intra-prediction neighbour availability of a macroblock (slice + picture bounds +
 * constrained_intra_pred already applied by the entropy kernel) */
#define MI_AV_LEFT 1
#define MI_AV_TOP 2
#define MI_AV_TOPLEFT 4
#define MI_AV_TOPRIGHT 8

typedef struct __attribute__((aligned(16))) {
    uint8_t type;     /* MBT_* */
    uint8_t t8x8;     /* transform_size_8x8_flag */
    uint8_t qp;       /* QP_Y (0 for I_PCM, as deblocking wants it) */
    uint8_t qpc[2];   /* QP_C for Cb, Cr */
    uint8_t cbp;      /* luma bits 0-3, chroma bits 4-5 */
    uint8_t chroma_mode;
    uint8_t i16mode;
    uint16_t nzmask;  /* 4x4 luma blocks (raster) with non-zero coefficients; 8x8 blocks replicated */
    uint8_t avail;    /* MI_AV_* */
    uint8_t dbf_idc;  /* disable_deblocking_filter_idc of the slice */
    int8_t alpha_off, beta_off; /* FilterOffsetA / FilterOffsetB */
    uint16_t slice_in_pic;      /* slice ordinal inside the picture (deblock idc 2, intra availability) */
    int8_t ipm[16];   /* Intra4x4/8x8PredMode per 4x4 block, raster; inter macroblocks: [0..3] = ref_idx_l1 per 8x8 (B slices), [4] = mb_type as
                       * coded (Tables 7-13 / 7-14), [5..8] = sub_mb_type per 8x8 (P_8x8 / B_8x8), rest 0 */
    int8_t ref[4];    /* ref_idx_l0 per 8x8 */
    int16_t refslot[4]; /* frame-pool slot of the referenced picture per 8x8 (-1 none) */
    uint32_t slice_idx; /* index into SliceDesc[] (weighted prediction tables) */
    int16_t mv[16][2];  /* final motion vectors per 4x4 block, raster, quarter-sample units */
    uint32_t coef_off;  /* first 32-byte block of this macroblock in the coefficient pool */
    uint32_t coef_mask; /* bit j: staging block j is present (packed in ascending order); I_PCM: 0xFFF = 384 sample bytes */
    int16_t refslot1[4]; /* list 1: frame-pool slot per 8x8, -1 = the quadrant does not use list 1 (always -1 outside B slices);
                          * the list-1 vectors of the pictures that have B slices live in a second array, MbMv1 */
} MbRec; /* 128 bytes */
#define MBREC_REF1(r) ((r)->ipm) /* ref_idx_l1 per 8x8 of an inter macroblock */

typedef struct __attribute__((aligned(16))) {
    int16_t mv[16][2];
} MbMv1; /* list-1 motion vectors per 4x4 block: one per MbRec, same index */

/* Motion a picture leaves behind for the direct prediction of later B pictures (8.4.1.2.1): per 4x4 block the vector of
 * the list the co-located block uses (list 0 if it uses it, otherwise list 1), per 8x8 the reference index and the frame
 * slot of the picture it points to (-1: intra).  One array per frame slot, written by k_dbprep. */
typedef struct __attribute__((aligned(16))) {
    int16_t mv[16][2];
    int16_t refslot[4];
    int8_t ref[4];
    uint8_t pad[4];
} ColRec; /* 80 bytes */

/* What K5 needs of a macroblock, worked out by k_dbprep for every macroblock of a batch at once (no dependencies, fully
 * parallel) so that the deblocking kernels -- a serial dependency chain per picture -- carry none of it: the 32 boundary
 * strengths (8.7.2.1) and, per colour plane, alpha / beta and the tC0 rows (8.7.2.2, Tables 8-16 / 8-17) of the three QP
 * averages a macroblock's edges use: its left edge, its inner edges, its top edge. */
typedef struct __attribute__((aligned(16))) {
    uint8_t bs[4][2][4];      /* [segment: rows (vertical edges) / columns (horizontal edges) 4 seg .. 4 seg + 3][direction: 0 vertical edges, 1 horizontal][edge]:
                               * what a K5 lane needs -- the strengths of the four edges crossing its line(s) -- is one dword per direction */
    struct {
        uint8_t ab[6];        /* alpha, beta of the left edge | of the inner edges | of the top edge */
        uint8_t tc[3][3];     /* tC0 for bS 1..3 of the left edge | inner edges | top edge */
        uint8_t pad;
    } pl[3];                  /* Y, Cb, Cr */
} DbPrm; /* 80 bytes */

typedef struct {
    uint32_t rbsp_off;     /* byte offset of the slice RBSP in the bitstream buffer */
    uint32_t rbsp_size;
    uint32_t data_bit_off; /* slice_data() bit offset inside the RBSP */
    uint32_t stop_bit;     /* bit position of rbsp_stop_one_bit (more_rbsp_data() for CAVLC) */
    uint32_t pic_idx;
    uint32_t first_mb;
    /* Macroblock range this slice's wavefront is responsible for: [fill_from, end_mb).  It decodes from first_mb and may
     * not pass end_mb (the next slice of the picture, or the picture size); whatever it does not decode inside the range
     * -- a gap in front of the slice, the rest after an error or an early end -- gets all-zero records (type MBT_NONE),
     * so that the reconstruction kernels never see stale or uninitialised records. */
    uint32_t fill_from, end_mb;
    uint8_t slice_type;    /* 0 P, 1 B, 2 I */
    uint8_t cabac_init_idc, slice_qp, num_ref_idx_active;
    int8_t alpha_off, beta_off;
    uint8_t dbf_idc, wp_flag;
    uint16_t slice_in_pic;
    uint8_t luma_log2_denom, chroma_log2_denom;
    int16_t ref_slot[MI_MAX_REFS];
    int16_t wp_lw[MI_MAX_REFS], wp_lo[MI_MAX_REFS];
    int16_t wp_cw[MI_MAX_REFS][2], wp_co[MI_MAX_REFS][2];
    uint32_t bext;         /* B slices: index of the slice's BSliceExt */
} SliceDesc;

/* What a B slice needs on top of its SliceDesc (host-built, 8.2.4.2.3 / 8.4.1.2 / 8.4.2.3) */
typedef struct {
    int16_t ref_slot1[MI_MAX_REFS];  /* RefPicList1 as frame-pool slots */
    int16_t wp_lw1[MI_MAX_REFS], wp_lo1[MI_MAX_REFS];
    int16_t wp_cw1[MI_MAX_REFS][2], wp_co1[MI_MAX_REFS][2];
    int16_t implicit_w1[MI_MAX_REFS][MI_MAX_REFS]; /* [refIdxL0][refIdxL1] -> w1 of 8.4.2.3.1 (w0 = 64 - w1), -64..128 */
    int16_t dist_scale[MI_MAX_REFS]; /* temporal direct: DistScaleFactor per refIdxL0; 256 where the vector is copied (long-term / equal POC) */
    uint64_t col;                    /* ColRec array of RefPicList1[0] (device address; 0: no such picture) */
    uint8_t col_short;               /* RefPicList1[0] is a short-term reference picture (colZeroFlag) */
    uint8_t direct_spatial, direct_8x8_inference;
    uint8_t wp_mode;                 /* weighted_bipred_idc: 0 default, 1 explicit, 2 implicit */
    uint8_t num_ref_idx_l1_active;
    uint8_t pad[3];
} BSliceExt;

typedef struct {
    uint32_t stream;
    uint32_t slot;        /* frame-pool slot this picture is reconstructed into */
    uint32_t wmb, hmb;
    uint64_t mb_base;     /* first MbRec / coefficient block of this picture */
    uint32_t first_slice, n_slices;
    uint8_t cabac, t8x8_mode, cip, weighted_pred;
    int8_t cqp_off[2];
    uint8_t is_intra_only; /* all slices are I slices */
    uint8_t scaling_set;   /* index into DevTables.level_scale sets */
    uint32_t order;        /* ordinal of this picture within its stream in this batch */
    /* the stream's frame pool, copied here so that a kernel needs ONE descriptor load per picture (a workgroup of K4 lives
     * for one macroblock: three dependent scalar loads -- list, picture, pool -- were 8 % of its lifetime) */
    uint32_t n_slots;
    uint64_t pool_base;    /* device address of slot 0 */
    uint64_t slot_bytes;
    uint64_t col_out;      /* ColRec array of this picture's frame slot */
    uint8_t has_b;         /* the picture has B slices: MbMv1 records exist, K4 runs its two-list variant */
    uint8_t save_col;      /* a later B picture (or batch) may ask for this picture's motion: k_dbprep also writes its ColRec array */
    uint8_t fmo;           /* more than one slice group: sgmap_off is valid, the records are zeroed before the entropy kernels run */
    uint8_t mono;          /* chroma_format_idc 0: the entropy kernels parse no chroma syntax (h264/sps.go:226-243) */
    uint32_t sgmap_off;    /* byte offset of the picture's mbToSliceGroupMap (8.2.2.8, one byte per macroblock) in the bitstream buffer */
    uint32_t inv_wmb;      /* floor(2^32 / wmb) + 1: mby = mulhi(mb, inv_wmb) is exact for mb < 2^32 / wmb / wmb (wmb <= 512, hmb <= 320) */
    /* Where the picture's samples are in its frame slot.  A frame: pitch = 16 wmb, plane = 16 wmb x 16 hmb.  A field picture
     * (h264/slice.go:867-872 field_pic_flag / bottom_field_flag) is reconstructed IN PLACE into the rows of its parity of the frame's slot:
     * hmb counts the field's macroblock rows, pitch is twice the frame's, plane the FRAME's plane size, and the first row starts
     * (field == 2 ? pitch / 2 : 0) bytes into the luma plane, (field == 2 ? pitch / 4 : 0) into each chroma plane. */
    uint32_t pitch;        /* luma bytes from one row of the picture to the next (chroma: half) */
    uint32_t plane;        /* offset of the Cb plane from the slot's first byte = luma bytes of the frame (Cr: plane * 5 / 4) */
    uint8_t field;         /* 0 frame picture, 1 top field, 2 bottom field */
    uint8_t pad2[7];
} PicDesc;
/* In field pictures a reference "slot" (SliceDesc::ref_slot, BSliceExt::ref_slot1, MbRec::refslot / refslot1, ColRec::refslot) names a FIELD:
 * the frame slot in the low bits and the field's parity in bit 14 (frame pictures never set it: they predict from whole frames). */
#define MI_REF_PARITY 0x4000
#define MI_REF_SLOT(r) ((r) & 0x3FFF)

typedef struct {
    uint64_t base;       /* device address of slot 0 */
    uint64_t slot_bytes; /* bytes per slot (Y + Cb + Cr) */
    uint32_t w, h;       /* coded luma size of the active sequence (kernels take the geometry of a picture from its PicDesc) */
    uint32_t n_slots;    /* slots in this pool: reference slots in MbRecs are clamped to it */
    uint32_t pad;
} FramePool;

/* LevelScale(m,i,j) of 8.5.9 for one PPS: [list][qp%6][raster position] */
typedef struct {
    uint16_t ls4[6][6][16];
    uint16_t ls8[2][6][64];
} ScalingSet;

#define MI_MAX_SCALING_SETS 8
/* CAVLC code tables in compact form (they live in LDS while a CAVLC slice is decoded): a code word is looked up by the number of its leading
 * zeros -- capped at the longest code of the table, L -- and the S bits behind its first one: entry [min(clz, L) << S | next S bits] =
 * len << 8 | value (value: total_coeff << 2 | trailing_ones for coeff_token), 0: no such code.  mi_api.cpp derives the tables from direct-indexed
 * ones and checks every possible window against those (h264mi_internal_vlc_selftest). */
#define MI_VLC_CT_S 3                      /* coeff_token: suffix bits */
#define MI_VLC_CT0_L 16
#define MI_VLC_CT1_L 14
#define MI_VLC_CT2_L 10
#define MI_VLC_CDC_L 8
#define MI_VLC_CDC_S 2
#define MI_VLC_TZ_L 9
#define MI_VLC_TZ_S 2
#define MI_VLC_CT0 0                                              /* 0 <= nC < 2 */
#define MI_VLC_CT1 (MI_VLC_CT0 + ((MI_VLC_CT0_L + 1) << MI_VLC_CT_S)) /* 2 <= nC < 4 */
#define MI_VLC_CT2 (MI_VLC_CT1 + ((MI_VLC_CT1_L + 1) << MI_VLC_CT_S)) /* 4 <= nC < 8 */
#define MI_VLC_CT3 (MI_VLC_CT2 + ((MI_VLC_CT2_L + 1) << MI_VLC_CT_S)) /* 8 <= nC: 6-bit fixed length, direct */
#define MI_VLC_CDC (MI_VLC_CT3 + 64)                              /* chroma DC coeff_token (nC = -1) */
#define MI_VLC_TZ (MI_VLC_CDC + ((MI_VLC_CDC_L + 1) << MI_VLC_CDC_S)) /* total_zeros, tzVlcIndex 1..15 */
#define MI_VLC_TZ_STRIDE ((MI_VLC_TZ_L + 1) << MI_VLC_TZ_S)
#define MI_VLC_CDCTZ (MI_VLC_TZ + 15 * MI_VLC_TZ_STRIDE)         /* chroma DC total_zeros: [3][8], 3 bits direct */
#define MI_VLC_RUN (MI_VLC_CDCTZ + 24)                            /* run_before, zerosLeft 1..6: [6][8], 3 bits direct (zerosLeft > 6: mi_run_before_long) */
#define MI_VLC_N ((MI_VLC_RUN + 48 + 3) & ~3)
/* index of a window (next 32 stream bits, MSB first) in a compact table of longest code L and S suffix bits */
#define MI_VLC_INDEX(w, lz, L, S) ((((lz) < (L) ? (lz) : (L)) << (S)) | ((uint32_t)((w) << (((lz) < (L) ? (lz) : (L)) + 1)) >> (32 - (S))))
/* run_before for zerosLeft > 6 (Table 9-10, last column: 111 .. 001 = 0 .. 6, then 0001 = 7, 00001 = 8, ...): len << 8 | run, 0 = no such code */
#define MI_RUN_BEFORE_LONG(w, lz) (((w) >> 29) ? (3u << 8 | (7u - ((w) >> 29))) : ((lz) <= 10 ? ((uint32_t)((lz) + 1) << 8 | (uint32_t)((lz) + 4)) : 0u))

typedef struct {
    uint8_t range_lps[64][4]; /* Table 9-44 */
    uint8_t trans_lps[64];    /* Table 9-45 */
    uint8_t ctx_init[4][52][464]; /* pStateIdx | valMPS << 6 for every (table set, SliceQPY, ctxIdx): 9.3.1.1 */
    uint8_t sig8x8[64], last8x8[64];
    uint8_t sig8x8_field[64]; /* Table 9-43, field-coded 8x8 blocks (field pictures) */
    uint8_t zigzag4[16], zigzag8[64];
    uint8_t fieldscan4[16], fieldscan8[64]; /* Tables 8-12 / 8-13, field scan: what a field picture's blocks are scanned in */
    uint8_t me_intra[64], me_inter[64]; /* Table 9-4: [0..47] ChromaArrayType 1 / 2, [48..63] ChromaArrayType 0 / 3 */
    uint8_t alpha[52], beta[52], tc0[52][4];
    uint8_t qpc[52];
    uint16_t vlc_c[MI_VLC_N]; /* the compact CAVLC tables (see MI_VLC_*) */
    ScalingSet scaling[MI_MAX_SCALING_SETS];
} DevTables;

#endif

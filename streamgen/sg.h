/*
 * streamgen/sg.h -- synthetic H.264 Annex-B stream generator (a small closed-loop encoder).
 *
 * Purpose: there is no network, no sample .h264 and no third-party encoder in the build image
 * (SURVEY.md fact 5), so tests and bench.py synthesise their inputs here.  The generator emits
 * (a) a conforming Annex-B byte stream and (b) its own reconstruction of every frame, produced by
 * an implementation of prediction / transform / deblocking written independently of both the
 * oracle (oracle/) and the product (h264decode_amd/).  Round-trip tests require
 * decoder output == generator reconstruction, bit for bit.
 *
 * This is input synthesis, not part of the decode product and not the oracle.
 */
#ifndef SG_H
#define SG_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int width, height;      /* display size; coded size is rounded up to 16, cropping signalled in the SPS */
    int frames;
    int profile_idc;        /* 66 Baseline, 77 Main, 100 High */
    int cabac;              /* entropy_coding_mode_flag */
    int qp;                 /* base QP (pic_init_qp); */
    int qp_jitter;          /* per-MB |mb_qp_delta| up to this value (0 = constant QP) */
    int idr_period;         /* 1 = all IDR; N = IDR every N frames, P in between; 0 = only the first frame is IDR */
    int slices;             /* slices per picture (split by MB rows as evenly as possible) */
    int transform8x8;       /* 0 off; 1 = High 8x8 transform + Intra8x8 enabled (needs profile 100) */
    int num_ref_frames;     /* 1..4 */
    int deblock_idc;        /* disable_deblocking_filter_idc: 0 on, 1 off, 2 on except slice edges */
    int alpha_off_div2, beta_off_div2;
    int cabac_init_idc;     /* 0..2, or -1: cycle per slice */
    int constrained_intra;  /* constrained_intra_pred_flag */
    int chroma_qp_offset;   /* chroma_qp_index_offset (second offset = same unless High: then +1 when transform8x8) */
    int pcm_permille;       /* probability (1/1000) of an I_PCM macroblock */
    int intra_in_p_permille;/* probability of an intra MB inside P pictures */
    int skip_permille;      /* probability of P_Skip */
    int sub8x8_permille;    /* probability of P_8x8 (with random sub-partitions) / 16x8 / 8x16 */
    int weighted_pred;      /* explicit weighted prediction in P slices */
    int scaling_matrix;     /* 0 flat, 1 = send default (non-flat) scaling lists in the SPS (High) */
    int noise;              /* amplitude of the uniform source noise */
    uint32_t seed;
    int long_start_code;    /* 1: 4-byte start codes everywhere; 0: 3-byte for non-parameter-set NALs */
    int poc_type;           /* 0, 1 or 2 */
    /* picture management stimulus (8.2.1, 8.2.4.3, 8.2.5.4); all 0 = sliding window, default lists, every picture a reference */
    int rplm;               /* 1: P slices carry random ref_pic_list_modification() commands (idc 0, 1 and, with long-term pictures, 2) */
    int mmco;               /* 1: reference P pictures carry random memory_management_control_operation scripts (1..6) */
    int idr_long_term;      /* 1: IDR pictures set long_term_reference_flag */
    int nonref_period;      /* N > 1: every N-th P picture is a non-reference picture (nal_ref_idc 0) */
    int slice_qp_delta;     /* d != 0: slice_qp_delta cycles through -d, 0, +d per slice */
    /* B pictures (Main / High): `bframes` non-reference B pictures between two anchors (display order I B B P -> coding
     * order I P B B); needs num_ref_frames >= 2 and forces pic_order_cnt_type 0 */
    int bframes;
    int direct_temporal;    /* 0: direct_spatial_mv_pred_flag = 1; 1: temporal direct */
    int weighted_bipred;    /* weighted_bipred_idc: 0 default average, 1 explicit, 2 implicit */
    int bskip_permille;     /* probability of B_Skip; B_Direct_16x16 gets half of it on top */
    int motion_x4, motion_y4; /* motion of the synthetic scene per frame in quarter samples (default 12, -8 = whole samples (3, -2));
                             * anything not a multiple of 4 makes fractional motion vectors the rule (6-tap interpolation) */
    int interlace_sps;      /* 1: frame_mbs_only_flag = 0 in the SPS (mb_adaptive_frame_field_flag = 0), every picture still a frame
                             * (field_pic_flag = 0): the syntax of a PAFF-capable stream that never uses a field picture */
    int fn_gap_period;      /* N > 0 (streams without B pictures / marking scripts): before every N-th picture after an IDR picture
                             * frame_num skips one or two values (8.2.5.2): the skipped frames enter the window as "non-existing"
                             * frames and the lists are re-ordered so that only real pictures are predicted from */
    int fn_gap_declared;    /* gaps_in_frame_num_value_allowed_flag of the SPS; 0 with fn_gap_period set = a stream that lost pictures */
    int b_pyramid;          /* with bframes >= 2: the middle B picture of a group is coded first, as a REFERENCE picture (nal_ref_idc 2);
                             * the other B pictures of the group may predict from it and take it as their co-located picture */
    /* Baseline extras (7.3.2.2, 8.2.2): slice_groups n >= 2 puts every picture's macroblocks into n slice groups by map type
     * fmo_type 0..6 (types 3..5: two groups, slice_group_change_cycle moves from picture to picture; 6: an explicit pseudo-random
     * map); `slices` then counts the slices PER GROUP.  aso: the slices of a picture leave in a shuffled order (arbitrary slice
     * order; needs more than one slice per picture to show) */
    int slice_groups, fmo_type, aso;
    /* PAFF: 1 = every frame is coded as two FIELD pictures (field_pic_flag = 1), top field first; 2 = bottom field first;
     * 3 = picture-adaptive: frame by frame either a frame picture or two field pictures, so that frames predict from
     * field-coded frames and fields from the fields of frame-coded ones.  The first field of an IDR frame is the IDR
     * picture, its second field a P or I field of the same frame_num; all other fields are P fields whose RefPicList0
     * alternates between fields of the same and of the opposite parity (8.2.4.2.5) and may start with the first field of
     * the same frame.  Forces interlace_sps; with cabac = 1 the field-coded blocks use the UNPINNED context values of sg_cabac_mn.c (ctxIdx 277..398,
     * 436..459: written down without the standard at hand); sliding-window marking (counted in frames, 8.2.5.3), list modification on field picture numbers
     * (rplm, P fields), marking scripts of operation 1 on single fields (mmco: a frame then lacks a field in later lists, and
     * a frame picture that finds no frame with both fields marked is coded as an I picture), non-reference frames
     * (nonref_period) and slice groups (a map unit is one macroblock of a field, 8.2.2.8) allowed, no long-term pictures.  bframes with field_pics 1 / 2 (not 3): every B frame is two non-reference B fields (spatial or
     * temporal direct prediction from the co-located field), their lists built from the anchors' fields by PicOrderCnt
     * (8.2.4.2.4) and the same alternation.  recon[] holds the woven frames. */
    int field_pics;
    /* d != 0: bottom_field_pic_order_in_frame_present_flag = 1 and every FRAME picture says where its bottom field sits relative to
     * its top field (delta_pic_order_cnt_bottom with pic_order_cnt_type 0, delta_pic_order_cnt[1] with type 1): BottomFieldOrderCnt =
     * TopFieldOrderCnt + d.  PicOrderCnt of a frame is the smaller of the two (8.2.1): a negative d moves every non-IDR picture d
     * earlier (IDR pictures keep 0 = Min(top, bottom): they send Max(d, 0); a negative d is taken as -1, the value of bottom-field-first material: anything lower would put the first pictures of a sequence before their IDR picture).  Ignored with pic_order_cnt_type 2. */
    int poc_bottom_delta;
    /* 1: chroma_format_idc 0 (monochrome; h264/sps.go:226-243 ChromaFormat; High profile only): no intra_chroma_pred_mode, coded_block_pattern by the
     * ChromaArrayType 0 column of Table 9-4 (CABAC: no chroma bins), no chroma residual, 256 samples per I_PCM macroblock, no chroma weights.  The
     * reconstruction carries chroma planes of 128 (what a decoder puts out for a 4:2:0 display). */
    int mono;
} sg_params;

void sg_default_params(sg_params *p);
/* Returns stream size in bytes (0 on failure / overflow).  recon (optional) receives every frame
 * at CODED size, I420 planar, back to back; recon_cap in bytes.  frame_sizes (optional, `frames`
 * entries) receives the byte size of each access unit. */
size_t sg_encode(const sg_params *p, uint8_t *stream, size_t stream_cap, uint8_t *recon, size_t recon_cap, uint32_t *frame_sizes);
/* source picture t of the synthetic sequence (coded size), for reference / PSNR */
void sg_source_frame(const sg_params *p, int t, uint8_t *dst);
const char *sg_last_error(void);
/* PicOrderCnt the generator intended for every picture of the last sg_encode() call (display order == coding order:
 * 2 * pictures since the last IDR / memory_management_control_operation 5).  Returns the number of pictures. */
int sg_last_pocs(int32_t *dst, int cap);
/* what the last sg_encode() call actually emitted: bit k = memory_management_control_operation k (1..6), bit 8/9/10 =
 * modification_of_pic_nums_idc 0/1/2, bit 11 = a long-term picture in an active reference list, bit 12 = non-reference
 * picture, bit 13 = slice_qp_delta != 0, bit 14 = pic_order_cnt_type 1 with delta_pic_order_cnt[0] != 0 */
uint32_t sg_last_features(void);

#ifdef __cplusplus
}
#endif
#endif

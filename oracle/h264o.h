/*
 * oracle/h264o.h -- CPU oracle: scalar C restatement of the H.264 Annex-B -> NAL -> slice ->
 * macroblock -> YCbCr path (ITU-T H.264 04/2017), frame-coded 4:2:0 8-bit, I, P and B slices,
 * CAVLC and CABAC, 4x4 and 8x8 transforms, in-loop deblocking.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / reported CPU baseline.
 * The product path (h264decode_amd/) never links, imports or calls anything in oracle/.
 *
 * PARITY STATUS: the reference (mrmod/h264decode, Go) cannot be built here (no Go toolchain,
 * un-vendored dependency github.com/mrmod/degolomb, no go.mod) and holds no golden vectors
 * (its only test, h264/server_test.go:8-16, does not compile and its fixture is git-ignored).
 * It also stops before residual decoding (h264/slice.go:599-828) and never produces a pixel.
 * Hence: "parity unpinned" against the reference itself.  What pins this oracle instead:
 *   - spec-table KATs (tests/test_tables.py), cross-checked against the reference's own
 *     table files where those are right (h264/rangeTabLPS.go, h264/stateTransxTab.go,
 *     h264/bit_reader.go:67-134) -- fixtures in tests/golden/;
 *   - an independent second implementation of reconstruction inside the stream generator
 *     (streamgen/), whose closed-loop reconstruction must equal this decoder's output;
 *   - a third-party x264 High-profile CABAC stream decoded to self-synchronisation.
 *
 * Reference counterparts are cited per function as h264/<file>.go:<line>.
 */
#ifndef H264O_H
#define H264O_H
#include <stddef.h>
#include <stdint.h>
#include "h264o_tables.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- bit reader over an RBSP (h264/bit_reader.go:11-17) ---------- */
typedef struct {
    const uint8_t *p;
    int64_t nbits; /* total bits available */
    int64_t pos;   /* bit cursor */
    int err;       /* set when reading past the end */
} h264o_br;

void h264o_br_init(h264o_br *b, const uint8_t *p, size_t nbytes);
uint32_t h264o_u(h264o_br *b, int n);   /* u(n), n<=32   h264/bit_reader.go:292-325 */
uint32_t h264o_peek(h264o_br *b, int n); /* n<=25, zero-padded past the end */
void h264o_skip(h264o_br *b, int n);
uint32_t h264o_ue(h264o_br *b);         /* h264/bit_reader.go:62-64 */
int32_t h264o_se(h264o_br *b);          /* h264/bit_reader.go:158-161 (buggy there for odd codeNum) */
uint32_t h264o_te(h264o_br *b, int range); /* h264/bit_reader.go:147-155 */
int h264o_more_rbsp_data(h264o_br *b);  /* h264/bit_reader.go:199-219 (destructive + inverted there) */

/* ---------- Annex B / NAL (h264/server.go:28-111, h264/nalUnit.go:75-131) ---------- */
typedef struct {
    int forbidden_zero_bit, nal_ref_idc, nal_unit_type;
    size_t offset; /* first byte of the NAL header inside the stream */
    size_t size;   /* NAL bytes incl. header, excl. start code and trailing zeros */
} h264o_nal;
/* Scan for 3- and 4-byte start codes (B.1).  Returns number of NALs written (<= cap). */
int h264o_annexb_scan(const uint8_t *buf, size_t len, h264o_nal *out, int cap);
/* Remove emulation_prevention_three_byte; returns RBSP length (payload after the 1-byte header). */
size_t h264o_nal_to_rbsp(const uint8_t *nal, size_t size, uint8_t *rbsp);

/* ---------- parameter sets (h264/sps.go:9-103,192-437; h264/pps.go:10-38,40-133) ---------- */
typedef struct {
    int valid;
    int profile_idc, constraint_set_flags, level_idc, seq_parameter_set_id;
    int chroma_format_idc, separate_colour_plane_flag, bit_depth_luma_minus8, bit_depth_chroma_minus8;
    int qpprime_y_zero_transform_bypass_flag, seq_scaling_matrix_present_flag;
    uint8_t scaling4x4[6][16]; /* resolved, zig-zag order */
    uint8_t scaling8x8[2][64];
    int log2_max_frame_num_minus4, pic_order_cnt_type, log2_max_pic_order_cnt_lsb_minus4;
    int delta_pic_order_always_zero_flag, offset_for_non_ref_pic, offset_for_top_to_bottom_field;
    int num_ref_frames_in_pic_order_cnt_cycle;
    int offset_for_ref_frame[256];
    int max_num_ref_frames, gaps_in_frame_num_value_allowed_flag;
    int pic_width_in_mbs_minus1, pic_height_in_map_units_minus1;
    int frame_mbs_only_flag, mb_adaptive_frame_field_flag, direct_8x8_inference_flag;
    int frame_cropping_flag, frame_crop_left_offset, frame_crop_right_offset, frame_crop_top_offset, frame_crop_bottom_offset;
    int vui_parameters_present_flag;
    int aspect_ratio_info_present_flag, aspect_ratio_idc, sar_width, sar_height;
    int overscan_info_present_flag, overscan_appropriate_flag;
    int video_signal_type_present_flag, video_format, video_full_range_flag, colour_description_present_flag;
    int colour_primaries, transfer_characteristics, matrix_coefficients;
    int chroma_loc_info_present_flag, chroma_sample_loc_type_top_field, chroma_sample_loc_type_bottom_field;
    int timing_info_present_flag;
    uint32_t num_units_in_tick, time_scale;
    int fixed_frame_rate_flag;
    int nal_hrd_parameters_present_flag, vcl_hrd_parameters_present_flag, low_delay_hrd_flag, pic_struct_present_flag;
    int cpb_cnt_minus1, bit_rate_scale, cpb_size_scale;
    int initial_cpb_removal_delay_length_minus1, cpb_removal_delay_length_minus1, dpb_output_delay_length_minus1, time_offset_length;
    int bitstream_restriction_flag, motion_vectors_over_pic_boundaries_flag, max_bytes_per_pic_denom, max_bits_per_mb_denom;
    int log2_max_mv_length_horizontal, log2_max_mv_length_vertical, max_num_reorder_frames, max_dec_frame_buffering;
} h264o_sps;

typedef struct {
    int valid;
    int pic_parameter_set_id, seq_parameter_set_id, entropy_coding_mode_flag;
    int bottom_field_pic_order_in_frame_present_flag, num_slice_groups_minus1;
    int num_ref_idx_l0_default_active_minus1, num_ref_idx_l1_default_active_minus1;
    int weighted_pred_flag, weighted_bipred_idc, pic_init_qp_minus26, pic_init_qs_minus26, chroma_qp_index_offset;
    int deblocking_filter_control_present_flag, constrained_intra_pred_flag, redundant_pic_cnt_present_flag;
    int transform_8x8_mode_flag, pic_scaling_matrix_present_flag, second_chroma_qp_index_offset;
    uint8_t scaling4x4[6][16]; /* resolved against the SPS, zig-zag order */
    uint8_t scaling8x8[2][64];
    /* slice groups (h264/pps.go:16-23, :57-80); slice_group_id[] of map type 6 is returned separately (h264o_parse_pps_ids) */
    int slice_group_map_type, run_length_minus1[8], top_left[8], bottom_right[8];
    int slice_group_change_direction_flag, slice_group_change_rate_minus1, pic_size_in_map_units_minus1;
} h264o_pps;

int h264o_parse_sps(const uint8_t *rbsp, size_t len, h264o_sps *sps);
int h264o_parse_pps(const uint8_t *rbsp, size_t len, const h264o_sps *sps_table, h264o_pps *pps);
/* the same, plus slice_group_id[] of a slice_group_map_type-6 PPS (one byte per map unit; *n_ids = 0 otherwise) */
int h264o_parse_pps_ids(const uint8_t *rbsp, size_t len, const h264o_sps *sps_table, h264o_pps *pps, uint8_t *ids, size_t cap, size_t *n_ids);
/* 8.2.2 (h264/slice.go:134-158, :457-552): mapUnitToSliceGroupMap / mbToSliceGroupMap of a picture, nextMbAddress */
int h264o_map_unit_to_slice_group_map(const h264o_sps *sps, const h264o_pps *pps, const uint8_t *ids, int slice_group_change_cycle, uint8_t *map);
int h264o_mb_to_slice_group_map(const h264o_sps *sps, const h264o_pps *pps, const uint8_t *ids, int slice_group_change_cycle, int field_pic, uint8_t *map);
int h264o_next_mb_address(const uint8_t *map, int n_mbs, int n);

/* ---------- slice header (h264/slice.go:23-75, 835-1048) ---------- */
typedef struct {
    int first_mb_in_slice, slice_type /* 0 P,1 B,2 I (mod 5) */, slice_type_raw, pic_parameter_set_id, colour_plane_id, frame_num;
    int field_pic_flag, bottom_field_flag, idr_pic_id, pic_order_cnt_lsb, delta_pic_order_cnt_bottom;
    int delta_pic_order_cnt[2], redundant_pic_cnt, direct_spatial_mv_pred_flag;
    int num_ref_idx_active_override_flag, num_ref_idx_l0_active_minus1, num_ref_idx_l1_active_minus1;
    int ref_pic_list_modification_flag_l0, n_rplm;
    int rplm_idc[66];
    int rplm_val[66];
    int luma_log2_weight_denom, chroma_log2_weight_denom;
    int luma_weight_l0_flag[32], luma_weight_l0[32], luma_offset_l0[32];
    int chroma_weight_l0_flag[32], chroma_weight_l0[32][2], chroma_offset_l0[32][2];
    int no_output_of_prior_pics_flag, long_term_reference_flag, adaptive_ref_pic_marking_mode_flag;
    int n_mmco;
    int mmco_op[66], mmco_arg1[66], mmco_arg2[66];
    int cabac_init_idc, slice_qp_delta, sp_for_switch_flag, slice_qs_delta;
    int disable_deblocking_filter_idc, slice_alpha_c0_offset_div2, slice_beta_offset_div2;
    int slice_group_change_cycle;
    /* B slices: list 1 (h264/slice.go:924-936 modification, :940-984 weights) */
    int ref_pic_list_modification_flag_l1, n_rplm1;
    int rplm1_idc[66];
    int rplm1_val[66];
    int luma_weight_l1_flag[32], luma_weight_l1[32], luma_offset_l1[32];
    int chroma_weight_l1_flag[32], chroma_weight_l1[32][2], chroma_offset_l1[32][2];
    /* derived */
    int nal_ref_idc, nal_unit_type, idr_flag, slice_qp_y;
    int64_t slice_data_bit_offset; /* bit position of slice_data() in the RBSP */
} h264o_slice_header;

int h264o_parse_slice_header(h264o_br *b, int nal_ref_idc, int nal_unit_type, const h264o_sps *sps_table,
                             const h264o_pps *pps_table, h264o_slice_header *sh);

/* ---------- whole-stream decode ---------- */
typedef struct {
    int width, height;             /* display (cropped) size */
    int coded_width, coded_height; /* multiples of 16 */
    int n_frames;
    int error; /* 0 ok */
    uint64_t n_mbs, n_bins, n_bits;
} h264o_stream_info;

typedef struct h264o_decoder h264o_decoder;
h264o_decoder *h264o_decoder_create(void);
void h264o_decoder_destroy(h264o_decoder *d);
/* crop != 0: frames are written at display size (I420: Y, Cb, Cr planes back to back);
 * crop == 0: at coded size.  out may be NULL to count frames only.  Frames are emitted in
 * DECODING order (with B pictures that is not the output order: h264o_last_pocs() gives the PicOrderCnt of each).
 * Returns 0 on success, negative on error. */
int h264o_decode_stream(h264o_decoder *d, const uint8_t *buf, size_t len, int crop, uint8_t *out, size_t out_cap,
                        h264o_stream_info *info);
/* Optional per-macroblock syntax trace of the LAST decoded frames (for syntax round-trip tests):
 * 8 int32 per MB: {mb_type_raw, cbp, qp, intra16/chroma mode, t8x8, mv0x, mv0y, ref0}.
 * Pass NULL to disable. cap in MBs across all frames. */
void h264o_set_mb_trace(h264o_decoder *d, int32_t *trace, size_t cap_mbs);
const char *h264o_last_error(h264o_decoder *d);
/* Test instrumentation of the last h264o_decode_stream() call.
 * features: bit k = memory_management_control_operation k executed (1..6), bit 8/9/10 = modification_of_pic_nums_idc
 * 0/1/2 applied, bit 11 = a long-term picture in a final RefPicList0, bit 12 = non-reference picture decoded, bit 13 =
 * slice_qp_delta != 0, bit 14 = pic_order_cnt_type 1 with delta_pic_order_cnt[0] != 0.
 * pocs: PicOrderCnt(CurrPic) (8.2.1; 0 after an operation 5) of every output picture, in output order. */
uint32_t h264o_last_features(h264o_decoder *d);
int h264o_last_pocs(h264o_decoder *d, int32_t *dst, int cap);

/* KAT helpers exported for tests */
int h264o_kat_ue(const uint8_t *bytes, size_t n, int count, uint32_t *out);
int h264o_kat_se(const uint8_t *bytes, size_t n, int count, int32_t *out);
/* Decode `count` CABAC decisions with a single context initialised to (pStateIdx,valMPS). */
int h264o_kat_cabac_bins(const uint8_t *bytes, size_t n, int pstate, int mps, int count, uint8_t *bins);
void h264o_kat_idct4x4(const int16_t *coef_raster, int16_t *res_out);
void h264o_kat_idct8x8(const int16_t *coef_raster, int16_t *res_out);
int h264o_sizeof_sps(void);
int h264o_sizeof_pps(void);
int h264o_sizeof_slice_header(void);

#ifdef __cplusplus
}
#endif
#endif

/*
 * oracle/h264o_ps.c -- SPS (7.3.2.1.1 + Annex E), PPS (7.3.2.2), slice header (7.3.3).
 * TEST INFRASTRUCTURE ONLY (see h264o.h).
 *
 * Field order follows h264/sps.go:192-437, h264/pps.go:40-133 and h264/slice.go:835-1048 where
 * those are right; divergences follow the spec and are marked.
 */
#include <stdlib.h>
#include <string.h>
#include "h264o.h"

static int is_high_profile(int p) {
    /* list identical to h264/sps.go:230 */
    return p == 100 || p == 110 || p == 122 || p == 244 || p == 44 || p == 83 || p == 86 || p == 118 || p == 128 || p == 138 || p == 139 ||
           p == 134 || p == 135;
}

/* 7.3.2.1.1.1 scaling_list().  Returns useDefaultScalingMatrixFlag.
 * (h264/sps.go:172-191 ignores that flag and writes into shared globals: Appendix A14.) */
static int parse_scaling_list(h264o_br *b, uint8_t *list, int size) {
    int last = 8, next = 8, use_default = 0;
    for (int j = 0; j < size; j++) {
        if (next != 0) {
            int delta = h264o_se(b);
            next = (last + delta + 256) % 256;
            use_default = (j == 0 && next == 0);
        }
        list[j] = (uint8_t)(next == 0 ? last : next);
        last = list[j];
    }
    return use_default;
}

static void flat_lists(uint8_t s4[6][16], uint8_t s8[2][64]) {
    memset(s4, 16, 6 * 16);
    memset(s8, 16, 2 * 64);
}

/* Table 7-2 fall-back rules.  fb4/fb8: lists to fall back to for indices 0,3 / 6,7
 * (rule A: the Default_* lists; rule B: the SPS lists). */
static void parse_scaling_matrix(h264o_br *b, int n_lists, uint8_t s4[6][16], uint8_t s8[2][64], const uint8_t *fb4_intra,
                                 const uint8_t *fb4_inter, const uint8_t *fb8_intra, const uint8_t *fb8_inter) {
    for (int i = 0; i < n_lists; i++) {
        int present = h264o_u(b, 1);
        if (i < 6) {
            if (present) {
                if (parse_scaling_list(b, s4[i], 16)) memcpy(s4[i], i < 3 ? h264o_default4x4_intra : h264o_default4x4_inter, 16);
            } else if (i == 0)
                memcpy(s4[0], fb4_intra, 16);
            else if (i == 3)
                memcpy(s4[3], fb4_inter, 16);
            else
                memcpy(s4[i], s4[i - 1], 16);
        } else {
            int k = i - 6;
            if (k >= 2) { /* 4:4:4 Cb/Cr 8x8 lists: parse and drop (out of scope) */
                uint8_t tmp[64];
                if (present) parse_scaling_list(b, tmp, 64);
                continue;
            }
            if (present) {
                if (parse_scaling_list(b, s8[k], 64)) memcpy(s8[k], k == 0 ? h264o_default8x8_intra : h264o_default8x8_inter, 64);
            } else
                memcpy(s8[k], k == 0 ? fb8_intra : fb8_inter, 64);
        }
    }
}

static void parse_hrd(h264o_br *b, h264o_sps *s) {
    /* E.1.2.  The length fields come AFTER the SchedSelIdx loop (h264/sps.go:211-214 reads
     * them inside it: Appendix A13). */
    s->cpb_cnt_minus1 = h264o_ue(b);
    s->bit_rate_scale = h264o_u(b, 4);
    s->cpb_size_scale = h264o_u(b, 4);
    for (int i = 0; i <= s->cpb_cnt_minus1 && i < 32; i++) {
        h264o_ue(b); /* bit_rate_value_minus1 */
        h264o_ue(b); /* cpb_size_value_minus1 */
        h264o_u(b, 1); /* cbr_flag */
    }
    s->initial_cpb_removal_delay_length_minus1 = h264o_u(b, 5);
    s->cpb_removal_delay_length_minus1 = h264o_u(b, 5);
    s->dpb_output_delay_length_minus1 = h264o_u(b, 5);
    s->time_offset_length = h264o_u(b, 5);
}

int h264o_parse_sps(const uint8_t *rbsp, size_t len, h264o_sps *s) {
    h264o_br br, *b = &br;
    h264o_br_init(b, rbsp, len);
    memset(s, 0, sizeof(*s));
    s->profile_idc = h264o_u(b, 8);
    s->constraint_set_flags = h264o_u(b, 8); /* 6 flags + reserved_zero_2bits */
    s->level_idc = h264o_u(b, 8);
    s->seq_parameter_set_id = h264o_ue(b);
    s->chroma_format_idc = 1;
    flat_lists(s->scaling4x4, s->scaling8x8);
    /* chroma_format_idc is only present for the High-family profiles (h264/sps.go:228 reads it
     * unconditionally: Appendix A11). */
    if (is_high_profile(s->profile_idc)) {
        s->chroma_format_idc = h264o_ue(b);
        if (s->chroma_format_idc == 3) s->separate_colour_plane_flag = h264o_u(b, 1);
        s->bit_depth_luma_minus8 = h264o_ue(b);
        s->bit_depth_chroma_minus8 = h264o_ue(b);
        s->qpprime_y_zero_transform_bypass_flag = h264o_u(b, 1);
        s->seq_scaling_matrix_present_flag = h264o_u(b, 1);
        if (s->seq_scaling_matrix_present_flag)
            parse_scaling_matrix(b, s->chroma_format_idc != 3 ? 8 : 12, s->scaling4x4, s->scaling8x8, h264o_default4x4_intra,
                                 h264o_default4x4_inter, h264o_default8x8_intra, h264o_default8x8_inter);
    }
    s->log2_max_frame_num_minus4 = h264o_ue(b);
    s->pic_order_cnt_type = h264o_ue(b);
    if (s->pic_order_cnt_type == 0)
        s->log2_max_pic_order_cnt_lsb_minus4 = h264o_ue(b);
    else if (s->pic_order_cnt_type == 1) {
        s->delta_pic_order_always_zero_flag = h264o_u(b, 1);
        s->offset_for_non_ref_pic = h264o_se(b);
        s->offset_for_top_to_bottom_field = h264o_se(b);
        s->num_ref_frames_in_pic_order_cnt_cycle = h264o_ue(b);
        if (s->num_ref_frames_in_pic_order_cnt_cycle > 255) return -1;
        for (int i = 0; i < s->num_ref_frames_in_pic_order_cnt_cycle; i++) s->offset_for_ref_frame[i] = h264o_se(b);
    }
    s->max_num_ref_frames = h264o_ue(b);
    s->gaps_in_frame_num_value_allowed_flag = h264o_u(b, 1);
    s->pic_width_in_mbs_minus1 = h264o_ue(b);
    s->pic_height_in_map_units_minus1 = h264o_ue(b);
    s->frame_mbs_only_flag = h264o_u(b, 1);
    if (!s->frame_mbs_only_flag) s->mb_adaptive_frame_field_flag = h264o_u(b, 1);
    s->direct_8x8_inference_flag = h264o_u(b, 1);
    s->frame_cropping_flag = h264o_u(b, 1);
    if (s->frame_cropping_flag) {
        s->frame_crop_left_offset = h264o_ue(b);
        s->frame_crop_right_offset = h264o_ue(b);
        s->frame_crop_top_offset = h264o_ue(b);
        s->frame_crop_bottom_offset = h264o_ue(b);
    }
    s->vui_parameters_present_flag = h264o_u(b, 1);
    if (s->vui_parameters_present_flag) {
        s->aspect_ratio_info_present_flag = h264o_u(b, 1);
        if (s->aspect_ratio_info_present_flag) {
            s->aspect_ratio_idc = h264o_u(b, 8);
            if (s->aspect_ratio_idc == 255) { /* Extended_SAR (h264/sps.go:346 uses 999: Appendix A12) */
                s->sar_width = h264o_u(b, 16);
                s->sar_height = h264o_u(b, 16);
            }
        }
        s->overscan_info_present_flag = h264o_u(b, 1);
        if (s->overscan_info_present_flag) s->overscan_appropriate_flag = h264o_u(b, 1);
        s->video_signal_type_present_flag = h264o_u(b, 1);
        if (s->video_signal_type_present_flag) {
            s->video_format = h264o_u(b, 3);
            s->video_full_range_flag = h264o_u(b, 1);
            s->colour_description_present_flag = h264o_u(b, 1);
            if (s->colour_description_present_flag) {
                s->colour_primaries = h264o_u(b, 8);
                s->transfer_characteristics = h264o_u(b, 8);
                s->matrix_coefficients = h264o_u(b, 8);
            }
        }
        s->chroma_loc_info_present_flag = h264o_u(b, 1);
        if (s->chroma_loc_info_present_flag) {
            s->chroma_sample_loc_type_top_field = h264o_ue(b);
            s->chroma_sample_loc_type_bottom_field = h264o_ue(b);
        }
        s->timing_info_present_flag = h264o_u(b, 1);
        if (s->timing_info_present_flag) {
            s->num_units_in_tick = h264o_u(b, 32);
            s->time_scale = h264o_u(b, 32);
            s->fixed_frame_rate_flag = h264o_u(b, 1);
        }
        s->nal_hrd_parameters_present_flag = h264o_u(b, 1);
        if (s->nal_hrd_parameters_present_flag) parse_hrd(b, s);
        s->vcl_hrd_parameters_present_flag = h264o_u(b, 1);
        if (s->vcl_hrd_parameters_present_flag) parse_hrd(b, s);
        if (s->nal_hrd_parameters_present_flag || s->vcl_hrd_parameters_present_flag) s->low_delay_hrd_flag = h264o_u(b, 1);
        s->pic_struct_present_flag = h264o_u(b, 1);
        s->bitstream_restriction_flag = h264o_u(b, 1);
        if (s->bitstream_restriction_flag) {
            s->motion_vectors_over_pic_boundaries_flag = h264o_u(b, 1);
            s->max_bytes_per_pic_denom = h264o_ue(b);
            s->max_bits_per_mb_denom = h264o_ue(b);
            s->log2_max_mv_length_horizontal = h264o_ue(b);
            s->log2_max_mv_length_vertical = h264o_ue(b);
            s->max_num_reorder_frames = h264o_ue(b);
            s->max_dec_frame_buffering = h264o_ue(b);
        }
    }
    if (b->err) return -1;
    if (s->seq_parameter_set_id > 31) return -1;
    s->valid = 1;
    return 0;
}

int h264o_parse_pps(const uint8_t *rbsp, size_t len, const h264o_sps *sps_table, h264o_pps *p) {
    return h264o_parse_pps_ids(rbsp, len, sps_table, p, NULL, 0, NULL);
}

static int bits_for(unsigned v) { /* Ceil(Log2(v)) */
    int n = 0;
    while ((1u << n) < v) n++;
    return n;
}

int h264o_parse_pps_ids(const uint8_t *rbsp, size_t len, const h264o_sps *sps_table, h264o_pps *p, uint8_t *ids, size_t cap, size_t *n_ids) {
    h264o_br br, *b = &br;
    h264o_br_init(b, rbsp, len);
    memset(p, 0, sizeof(*p));
    if (n_ids) *n_ids = 0;
    p->pic_parameter_set_id = h264o_ue(b);
    p->seq_parameter_set_id = h264o_ue(b);
    if (p->pic_parameter_set_id > 255 || p->seq_parameter_set_id > 31) return -1;
    const h264o_sps *s = &sps_table[p->seq_parameter_set_id];
    if (!s->valid) return -2;
    p->entropy_coding_mode_flag = h264o_u(b, 1);
    p->bottom_field_pic_order_in_frame_present_flag = h264o_u(b, 1);
    p->num_slice_groups_minus1 = h264o_ue(b);
    if (p->num_slice_groups_minus1 > 7) return -3;
    if (p->num_slice_groups_minus1 > 0) { /* 7.3.2.2; h264/pps.go:57-80 */
        int ng = p->num_slice_groups_minus1 + 1, wmb = s->pic_width_in_mbs_minus1 + 1;
        unsigned units = (unsigned)wmb * (unsigned)(s->pic_height_in_map_units_minus1 + 1);
        p->slice_group_map_type = h264o_ue(b);
        switch (p->slice_group_map_type) {
        case 0:
            for (int g = 0; g < ng; g++) p->run_length_minus1[g] = h264o_ue(b);
            break;
        case 1: break;
        case 2:
            for (int g = 0; g < ng - 1; g++) {
                p->top_left[g] = h264o_ue(b);
                p->bottom_right[g] = h264o_ue(b);
                if (p->top_left[g] > p->bottom_right[g] || (unsigned)p->bottom_right[g] >= units || p->top_left[g] % wmb > p->bottom_right[g] % wmb) return -3;
            }
            break;
        case 3:
        case 4:
        case 5:
            p->slice_group_change_direction_flag = h264o_u(b, 1);
            p->slice_group_change_rate_minus1 = h264o_ue(b);
            if (ng != 2 || (unsigned)p->slice_group_change_rate_minus1 >= units) return -3;
            break;
        case 6: {
            p->pic_size_in_map_units_minus1 = h264o_ue(b);
            if ((unsigned)p->pic_size_in_map_units_minus1 + 1 != units) return -3;
            int nb = bits_for((unsigned)ng);
            for (unsigned i = 0; i < units && !b->err; i++) {
                unsigned v = h264o_u(b, nb);
                if (v >= (unsigned)ng) return -3;
                if (ids && i < cap) ids[i] = (uint8_t)v;
            }
            if (n_ids) *n_ids = units;
            break;
        }
        default: return -3;
        }
    }
    p->num_ref_idx_l0_default_active_minus1 = h264o_ue(b);
    p->num_ref_idx_l1_default_active_minus1 = h264o_ue(b);
    p->weighted_pred_flag = h264o_u(b, 1);
    p->weighted_bipred_idc = h264o_u(b, 2);
    p->pic_init_qp_minus26 = h264o_se(b);
    p->pic_init_qs_minus26 = h264o_se(b);
    p->chroma_qp_index_offset = h264o_se(b);
    p->deblocking_filter_control_present_flag = h264o_u(b, 1);
    p->constrained_intra_pred_flag = h264o_u(b, 1);
    p->redundant_pic_cnt_present_flag = h264o_u(b, 1);
    p->second_chroma_qp_index_offset = p->chroma_qp_index_offset;
    memcpy(p->scaling4x4, s->scaling4x4, sizeof(p->scaling4x4));
    memcpy(p->scaling8x8, s->scaling8x8, sizeof(p->scaling8x8));
    /* more_rbsp_data() decides whether the High tail is present (h264/pps.go:93 uses
     * "bytes left": Appendix A15; second_chroma_qp_index_offset is unconditional in the tail:
     * Appendix A16). */
    if (h264o_more_rbsp_data(b)) {
        p->transform_8x8_mode_flag = h264o_u(b, 1);
        p->pic_scaling_matrix_present_flag = h264o_u(b, 1);
        if (p->pic_scaling_matrix_present_flag) {
            int n = 6 + ((s->chroma_format_idc != 3) ? 2 : 6) * p->transform_8x8_mode_flag;
            if (s->seq_scaling_matrix_present_flag) /* fall-back rule B */
                parse_scaling_matrix(b, n, p->scaling4x4, p->scaling8x8, s->scaling4x4[0], s->scaling4x4[3], s->scaling8x8[0], s->scaling8x8[1]);
            else /* fall-back rule A */
                parse_scaling_matrix(b, n, p->scaling4x4, p->scaling8x8, h264o_default4x4_intra, h264o_default4x4_inter, h264o_default8x8_intra,
                                     h264o_default8x8_inter);
        }
        p->second_chroma_qp_index_offset = h264o_se(b);
    }
    if (b->err) return -1;
    p->valid = 1;
    return 0;
}

/* 7.3.3 slice_header().  h264/slice.go:857-1032 with: frame_num actually read (A18), override
 * flag for P/SP/B (A19), a terminating MMCO loop (A20). */
int h264o_parse_slice_header(h264o_br *b, int nal_ref_idc, int nal_unit_type, const h264o_sps *sps_table, const h264o_pps *pps_table,
                             h264o_slice_header *sh) {
    memset(sh, 0, sizeof(*sh));
    sh->nal_ref_idc = nal_ref_idc;
    sh->nal_unit_type = nal_unit_type;
    sh->idr_flag = nal_unit_type == 5;
    sh->first_mb_in_slice = h264o_ue(b);
    sh->slice_type_raw = h264o_ue(b);
    if (sh->slice_type_raw > 9) return -1;
    sh->slice_type = sh->slice_type_raw % 5;
    sh->pic_parameter_set_id = h264o_ue(b);
    if (sh->pic_parameter_set_id > 255 || !pps_table[sh->pic_parameter_set_id].valid) return -2;
    const h264o_pps *p = &pps_table[sh->pic_parameter_set_id];
    const h264o_sps *s = &sps_table[p->seq_parameter_set_id];
    if (!s->valid) return -2;
    if (s->separate_colour_plane_flag) sh->colour_plane_id = h264o_u(b, 2);
    sh->frame_num = h264o_u(b, s->log2_max_frame_num_minus4 + 4);
    if (!s->frame_mbs_only_flag) {
        sh->field_pic_flag = h264o_u(b, 1);
        if (sh->field_pic_flag) sh->bottom_field_flag = h264o_u(b, 1);
    }
    if (sh->idr_flag) sh->idr_pic_id = h264o_ue(b);
    if (s->pic_order_cnt_type == 0) {
        sh->pic_order_cnt_lsb = h264o_u(b, s->log2_max_pic_order_cnt_lsb_minus4 + 4);
        if (p->bottom_field_pic_order_in_frame_present_flag && !sh->field_pic_flag) sh->delta_pic_order_cnt_bottom = h264o_se(b);
    }
    if (s->pic_order_cnt_type == 1 && !s->delta_pic_order_always_zero_flag) {
        sh->delta_pic_order_cnt[0] = h264o_se(b);
        if (p->bottom_field_pic_order_in_frame_present_flag && !sh->field_pic_flag) sh->delta_pic_order_cnt[1] = h264o_se(b);
    }
    if (p->redundant_pic_cnt_present_flag) sh->redundant_pic_cnt = h264o_ue(b);
    if (sh->slice_type == 1) sh->direct_spatial_mv_pred_flag = h264o_u(b, 1);
    sh->num_ref_idx_l0_active_minus1 = p->num_ref_idx_l0_default_active_minus1;
    sh->num_ref_idx_l1_active_minus1 = p->num_ref_idx_l1_default_active_minus1;
    if (sh->slice_type == 0 || sh->slice_type == 3 || sh->slice_type == 1) {
        sh->num_ref_idx_active_override_flag = h264o_u(b, 1);
        if (sh->num_ref_idx_active_override_flag) {
            sh->num_ref_idx_l0_active_minus1 = h264o_ue(b);
            if (sh->slice_type == 1) sh->num_ref_idx_l1_active_minus1 = h264o_ue(b);
        }
        if (sh->num_ref_idx_l0_active_minus1 > 31) return -1;
    }
    /* ref_pic_list_modification() 7.3.3.1 */
    if (sh->slice_type != 2 && sh->slice_type != 4) {
        sh->ref_pic_list_modification_flag_l0 = h264o_u(b, 1);
        if (sh->ref_pic_list_modification_flag_l0) {
            for (;;) {
                int idc = h264o_ue(b);
                if (idc == 3) break;
                if (idc > 3 || sh->n_rplm >= 66 || b->err) return -1;
                sh->rplm_idc[sh->n_rplm] = idc;
                sh->rplm_val[sh->n_rplm] = h264o_ue(b);
                sh->n_rplm++;
            }
        }
    }
    if (sh->slice_type == 1) { /* list 1 of B slices (h264/slice.go:924-936) */
        if (sh->num_ref_idx_l1_active_minus1 > 31) return -1;
        sh->ref_pic_list_modification_flag_l1 = h264o_u(b, 1);
        if (sh->ref_pic_list_modification_flag_l1) {
            for (;;) {
                int idc = h264o_ue(b);
                if (idc == 3) break;
                if (idc > 3 || sh->n_rplm1 >= 66 || b->err) return -1;
                sh->rplm1_idc[sh->n_rplm1] = idc;
                sh->rplm1_val[sh->n_rplm1] = h264o_ue(b);
                sh->n_rplm1++;
            }
        }
    }
    /* pred_weight_table() 7.3.3.2 */
    if ((p->weighted_pred_flag && (sh->slice_type == 0 || sh->slice_type == 3)) || (p->weighted_bipred_idc == 1 && sh->slice_type == 1)) {
        sh->luma_log2_weight_denom = h264o_ue(b);
        sh->chroma_log2_weight_denom = s->chroma_format_idc ? h264o_ue(b) : 0; /* ChromaArrayType != 0 (monochrome: no chroma weights, h264/sps.go:226-243) */
        if (sh->luma_log2_weight_denom > 7 || sh->chroma_log2_weight_denom > 7) return -1;
#define WP_BAD(v) ((v) < -128 || (v) > 127) /* 7.4.3.2: coded weights and offsets; the default weight 1 << denom is not coded */
        for (int i = 0; i <= sh->num_ref_idx_l0_active_minus1; i++) {
            sh->luma_weight_l0[i] = 1 << sh->luma_log2_weight_denom;
            sh->chroma_weight_l0[i][0] = sh->chroma_weight_l0[i][1] = 1 << sh->chroma_log2_weight_denom;
            sh->luma_weight_l0_flag[i] = h264o_u(b, 1);
            if (sh->luma_weight_l0_flag[i]) {
                sh->luma_weight_l0[i] = h264o_se(b);
                sh->luma_offset_l0[i] = h264o_se(b);
                if (WP_BAD(sh->luma_weight_l0[i]) || WP_BAD(sh->luma_offset_l0[i])) return -1;
            }
            sh->chroma_weight_l0_flag[i] = s->chroma_format_idc ? h264o_u(b, 1) : 0;
            if (sh->chroma_weight_l0_flag[i])
                for (int j = 0; j < 2; j++) {
                    sh->chroma_weight_l0[i][j] = h264o_se(b);
                    sh->chroma_offset_l0[i][j] = h264o_se(b);
                    if (WP_BAD(sh->chroma_weight_l0[i][j]) || WP_BAD(sh->chroma_offset_l0[i][j])) return -1;
                }
        }
        if (sh->slice_type == 1)
            for (int i = 0; i <= sh->num_ref_idx_l1_active_minus1; i++) {
                sh->luma_weight_l1[i] = 1 << sh->luma_log2_weight_denom;
                sh->chroma_weight_l1[i][0] = sh->chroma_weight_l1[i][1] = 1 << sh->chroma_log2_weight_denom;
                sh->luma_weight_l1_flag[i] = h264o_u(b, 1);
                if (sh->luma_weight_l1_flag[i]) {
                    sh->luma_weight_l1[i] = h264o_se(b);
                    sh->luma_offset_l1[i] = h264o_se(b);
                    if (WP_BAD(sh->luma_weight_l1[i]) || WP_BAD(sh->luma_offset_l1[i])) return -1;
                }
                sh->chroma_weight_l1_flag[i] = s->chroma_format_idc ? h264o_u(b, 1) : 0;
                if (sh->chroma_weight_l1_flag[i])
                    for (int j = 0; j < 2; j++) {
                        sh->chroma_weight_l1[i][j] = h264o_se(b);
                        sh->chroma_offset_l1[i][j] = h264o_se(b);
                        if (WP_BAD(sh->chroma_weight_l1[i][j]) || WP_BAD(sh->chroma_offset_l1[i][j])) return -1;
                    }
            }
    }
    /* dec_ref_pic_marking() 7.3.3.3 */
    if (nal_ref_idc != 0) {
        if (sh->idr_flag) {
            sh->no_output_of_prior_pics_flag = h264o_u(b, 1);
            sh->long_term_reference_flag = h264o_u(b, 1);
        } else {
            sh->adaptive_ref_pic_marking_mode_flag = h264o_u(b, 1);
            if (sh->adaptive_ref_pic_marking_mode_flag) {
                for (;;) {
                    int op = h264o_ue(b);
                    if (op == 0) break;
                    if (op > 6 || sh->n_mmco >= 66 || b->err) return -1;
                    sh->mmco_op[sh->n_mmco] = op;
                    if (op == 1 || op == 3) sh->mmco_arg1[sh->n_mmco] = h264o_ue(b); /* difference_of_pic_nums_minus1 */
                    if (op == 2) sh->mmco_arg1[sh->n_mmco] = h264o_ue(b);            /* long_term_pic_num */
                    if (op == 3 || op == 6) sh->mmco_arg2[sh->n_mmco] = h264o_ue(b); /* long_term_frame_idx */
                    if (op == 4) sh->mmco_arg1[sh->n_mmco] = h264o_ue(b);            /* max_long_term_frame_idx_plus1 */
                    sh->n_mmco++;
                }
            }
        }
    }
    if (p->entropy_coding_mode_flag && sh->slice_type != 2 && sh->slice_type != 4) {
        sh->cabac_init_idc = h264o_ue(b);
        if (sh->cabac_init_idc > 2) return -1;
    }
    sh->slice_qp_delta = h264o_se(b);
    if (sh->slice_type == 3 || sh->slice_type == 4) {
        if (sh->slice_type == 3) sh->sp_for_switch_flag = h264o_u(b, 1);
        sh->slice_qs_delta = h264o_se(b);
    }
    if (p->deblocking_filter_control_present_flag) {
        sh->disable_deblocking_filter_idc = h264o_ue(b);
        if (sh->disable_deblocking_filter_idc != 1) {
            sh->slice_alpha_c0_offset_div2 = h264o_se(b);
            sh->slice_beta_offset_div2 = h264o_se(b);
        }
    }
    if (p->num_slice_groups_minus1 > 0 && p->slice_group_map_type >= 3 && p->slice_group_map_type <= 5) {
        /* Ceil(Log2(PicSizeInMapUnits / SliceGroupChangeRate + 1)) bits with an exact quotient (7.4.3; h264/slice.go:1028-1031) */
        unsigned long long units = (unsigned long long)(s->pic_width_in_mbs_minus1 + 1) * (unsigned long long)(s->pic_height_in_map_units_minus1 + 1);
        unsigned long long rate = (unsigned long long)p->slice_group_change_rate_minus1 + 1;
        int n = 0;
        while (((1ull << n) - 1) * rate < units) n++;
        sh->slice_group_change_cycle = n ? (int)h264o_u(b, n) : 0;
        if ((unsigned long long)sh->slice_group_change_cycle > (units + rate - 1) / rate) return -1;
    }
    sh->slice_qp_y = 26 + p->pic_init_qp_minus26 + sh->slice_qp_delta; /* (7-30), h264/cabac.go:113 */
    sh->slice_data_bit_offset = b->pos;
    if (b->err) return -1;
    if (sh->slice_qp_y < 0 || sh->slice_qp_y > 51) return -1;
    return 0;
}

/* ------------------------------------------------------------------ slice groups 8.2.2 */
/* MapUnitToSliceGroupMap (h264/slice.go:457-529: types 0-2 only there).  map: PicSizeInMapUnits bytes. */
int h264o_map_unit_to_slice_group_map(const h264o_sps *s, const h264o_pps *p, const uint8_t *ids, int cycle, uint8_t *map) {
    int W = s->pic_width_in_mbs_minus1 + 1, H = s->pic_height_in_map_units_minus1 + 1, units = W * H, ng = p->num_slice_groups_minus1 + 1;
    if (ng == 1) {
        memset(map, 0, (size_t)units);
        return units;
    }
    int dir = p->slice_group_change_direction_flag, rate = p->slice_group_change_rate_minus1 + 1;
    long long g0 = (long long)cycle * rate; /* MapUnitsInSliceGroup0 (7-33) */
    if (g0 > units) g0 = units;
    long long upper_left = dir ? units - g0 : g0; /* (8-14) */
    switch (p->slice_group_map_type) {
    case 0: { /* 8.2.2.1 */
        int i = 0;
        while (i < units)
            for (int g = 0; g < ng && i < units; g++) {
                for (int j = 0; j <= p->run_length_minus1[g] && i + j < units; j++) map[i + j] = (uint8_t)g;
                i += p->run_length_minus1[g] + 1;
            }
        break;
    }
    case 1: /* 8.2.2.2 */
        for (int i = 0; i < units; i++) map[i] = (uint8_t)(((i % W) + (((i / W) * ng) / 2)) % ng);
        break;
    case 2: /* 8.2.2.3 */
        memset(map, ng - 1, (size_t)units);
        for (int g = ng - 2; g >= 0; g--)
            for (int y = p->top_left[g] / W; y <= p->bottom_right[g] / W; y++)
                for (int x = p->top_left[g] % W; x <= p->bottom_right[g] % W; x++) map[y * W + x] = (uint8_t)g;
        break;
    case 3: { /* 8.2.2.4 */
        memset(map, 1, (size_t)units);
        int x = (W - dir) / 2, y = (H - dir) / 2, lb = x, tb = y, rb = x, bb = y, xd = dir - 1, yd = dir;
        for (long long k = 0; k < g0;) {
            int vacant = map[y * W + x] == 1;
            if (vacant) map[y * W + x] = 0;
            if (xd == -1 && x == lb) {
                lb = lb - 1 < 0 ? 0 : lb - 1;
                x = lb, xd = 0, yd = 2 * dir - 1;
            } else if (xd == 1 && x == rb) {
                rb = rb + 1 > W - 1 ? W - 1 : rb + 1;
                x = rb, xd = 0, yd = 1 - 2 * dir;
            } else if (yd == -1 && y == tb) {
                tb = tb - 1 < 0 ? 0 : tb - 1;
                y = tb, xd = 1 - 2 * dir, yd = 0;
            } else if (yd == 1 && y == bb) {
                bb = bb + 1 > H - 1 ? H - 1 : bb + 1;
                y = bb, xd = 2 * dir - 1, yd = 0;
            } else
                x += xd, y += yd;
            k += vacant;
        }
        break;
    }
    case 4: /* 8.2.2.5 */
        for (int i = 0; i < units; i++) map[i] = (uint8_t)(i < upper_left ? dir : 1 - dir);
        break;
    case 5: { /* 8.2.2.6 */
        long long k = 0;
        for (int j = 0; j < W; j++)
            for (int i = 0; i < H; i++) map[i * W + j] = (uint8_t)(k++ < upper_left ? dir : 1 - dir);
        break;
    }
    case 6: /* 8.2.2.7 */
        if (!ids) return -1;
        memcpy(map, ids, (size_t)units);
        break;
    default: return -1;
    }
    return units;
}

/* MbToSliceGroupMap 8.2.2.8 (h264/slice.go:134-158).  map: PicSizeInMbs bytes; returns PicSizeInMbs. */
int h264o_mb_to_slice_group_map(const h264o_sps *s, const h264o_pps *p, const uint8_t *ids, int cycle, int field_pic, uint8_t *map) {
    int W = s->pic_width_in_mbs_minus1 + 1, units = W * (s->pic_height_in_map_units_minus1 + 1);
    if (s->frame_mbs_only_flag || field_pic) return h264o_map_unit_to_slice_group_map(s, p, ids, cycle, map);
    uint8_t *mu = (uint8_t *)malloc((size_t)units);
    if (h264o_map_unit_to_slice_group_map(s, p, ids, cycle, mu) < 0) {
        free(mu);
        return -1;
    }
    for (int i = 0; i < 2 * units; i++) map[i] = s->mb_adaptive_frame_field_flag ? mu[i / 2] : mu[(i / (2 * W)) * W + (i % W)];
    free(mu);
    return 2 * units;
}

/* nextMbAddress (8-17; h264/slice.go:530-552) */
int h264o_next_mb_address(const uint8_t *map, int n_mbs, int n) {
    int i = n + 1;
    while (i < n_mbs && map[i] != map[n]) i++;
    return i;
}

int h264o_sizeof_sps(void) { return (int)sizeof(h264o_sps); }
int h264o_sizeof_pps(void) { return (int)sizeof(h264o_pps); }
int h264o_sizeof_slice_header(void) { return (int)sizeof(h264o_slice_header); }

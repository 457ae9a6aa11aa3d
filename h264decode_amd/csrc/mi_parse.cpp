// h264decode_amd/csrc/mi_parse.cpp -- see mi_parse.hpp.
//
// Follows the syntax order of the reference where the reference is right and the spec elsewhere;
// each divergence is tagged with its SURVEY.md Appendix-A number.
#include "mi_parse.hpp"
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <vector>
#include "mi_tables.h"

namespace mi {

static thread_local char g_err[512];
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

// ---------------------------------------------------------------- BitReader
uint64_t BitReader::window(int64_t bitpos) const {
    int64_t byte = bitpos >> 3;
    uint64_t w = 0;
    int64_t nbytes = nbits_ >> 3;
    for (int i = 0; i < 9; i++) { // 72 bits gathered, then aligned
        uint64_t b = (byte + i < nbytes) ? p_[byte + i] : 0;
        if (i < 8)
            w = (w << 8) | b;
        else {
            int sh = static_cast<int>(bitpos & 7);
            if (sh) w = (w << sh) | (b >> (8 - sh));
        }
    }
    return w;
}
uint32_t BitReader::u(int n) {
    if (n == 0) return 0;
    uint64_t w = window(pos_);
    pos_ += n;
    return static_cast<uint32_t>(w >> (64 - n));
}
// ue(v): count leading zeros with one CLZ instead of a bit loop (h264/bit_reader.go:174-196).
uint32_t BitReader::ue() {
    uint64_t w = window(pos_);
    if (w == 0) { // > 63 leading zeros: malformed
        pos_ = nbits_ + 1;
        return 0;
    }
    int lz = __builtin_clzll(w);
    if (lz > 31) {
        pos_ = nbits_ + 1;
        return 0;
    }
    pos_ += 2 * lz + 1;
    return static_cast<uint32_t>((w >> (63 - 2 * lz)) - 1);
}
// se(v): 9.1.1 (the reference is off by one for odd codeNum, A3)
int32_t BitReader::se() {
    uint32_t k = ue();
    int32_t m = static_cast<int32_t>((k + 1) >> 1);
    return (k & 1) ? m : -m;
}
// more_rbsp_data(): non-destructive look for anything before the final stop bit (A29)
bool BitReader::more_rbsp_data() const {
    int64_t n = nbits_ >> 3;
    while (n > 0 && p_[n - 1] == 0) n--;
    if (n == 0) return false;
    int tz = __builtin_ctz(p_[n - 1]);
    int64_t stop_bit = (n - 1) * 8 + (7 - tz);
    return pos_ < stop_bit;
}

// ---------------------------------------------------------------- Annex B / NAL
// B.1: start code prefix 00 00 01 with optional leading zero bytes; trailing zeros dropped (A31).
int annexb_scan(const uint8_t *buf, size_t len, h264mi_nal *out, int cap, int *n) {
    int count = 0;
    size_t i = 0, nal_start = SIZE_MAX;
    auto emit = [&](size_t s, size_t e) {
        while (e > s && buf[e - 1] == 0) e--;
        if (e <= s) return;
        if (count < cap) {
            h264mi_nal &x = out[count];
            memset(&x, 0, sizeof(x));
            x.offset = static_cast<int64_t>(s);
            x.num_bytes = static_cast<int32_t>(e - s);
            x.forbidden_zero_bit = buf[s] >> 7;
            x.ref_idc = (buf[s] >> 5) & 3;
            x.type = buf[s] & 31;
            x.header_bytes = (x.type == 14 || x.type == 20 || x.type == 21) ? 4 : 1;
        }
        count++;
    };
    while (i + 2 < len) {
        // skip quickly: a start code needs buf[i+2] <= 1
        if (buf[i + 2] > 1) {
            i += 3;
            continue;
        }
        if (buf[i] == 0 && buf[i + 1] == 0 && buf[i + 2] == 1) {
            if (nal_start != SIZE_MAX) emit(nal_start, i);
            nal_start = i + 3;
            i += 3;
        } else
            i++;
    }
    if (nal_start != SIZE_MAX && nal_start < len) emit(nal_start, len);
    *n = count < cap ? count : cap;
    return count > cap ? H264MI_ECAPACITY : H264MI_OK;
}

// 7.4.1.1: drop emulation_prevention_three_byte
size_t unescape(const uint8_t *src, size_t n, uint8_t *dst) {
    size_t o = 0, i = 0;
    while (i < n) {
        // copy up to the next possible 00 00 03
        if (i + 2 < n && src[i] == 0 && src[i + 1] == 0 && src[i + 2] == 3) {
            dst[o++] = 0;
            dst[o++] = 0;
            i += 3;
        } else
            dst[o++] = src[i++];
    }
    return o;
}

// NewNalUnit (h264/nalUnit.go:75-131): header fields + RBSP
int nal_parse(const uint8_t *nal, size_t len, h264mi_nal *h, uint8_t *rbsp, size_t *rbsp_len) {
    if (!nal || len < 1 || !h) return H264MI_EINVAL;
    int64_t keep = h->offset;
    memset(h, 0, sizeof(*h));
    h->offset = keep;
    h->num_bytes = static_cast<int32_t>(len);
    h->forbidden_zero_bit = nal[0] >> 7;
    h->ref_idc = (nal[0] >> 5) & 3;
    h->type = nal[0] & 31;
    h->header_bytes = 1;
    if (h->type == 14 || h->type == 20 || h->type == 21) { // 7.3.1: 3 more header bytes (Annex G/H/J)
        if (len < 4) return H264MI_EBITSTREAM;
        if (h->type != 21)
            h->svc_extension_flag = nal[1] >> 7;
        else
            h->avc_3d_extension_flag = nal[1] >> 7;
        h->header_bytes = 4;
    }
    if (rbsp && rbsp_len) *rbsp_len = unescape(nal + h->header_bytes, len - h->header_bytes, rbsp);
    return H264MI_OK;
}

// ---------------------------------------------------------------- scaling lists
static bool scaling_list(BitReader &b, uint8_t *list, int size) { // 7.3.2.1.1.1; true = use default
    int last = 8, next = 8;
    bool use_default = false;
    for (int j = 0; j < size; j++) {
        if (next != 0) {
            next = (last + b.se() + 256) % 256;
            use_default = (j == 0 && next == 0);
        }
        list[j] = static_cast<uint8_t>(next == 0 ? last : next);
        last = list[j];
    }
    return use_default;
}
struct Fallback {
    const uint8_t *i4, *p4, *i8, *p8;
};
static void scaling_matrix(BitReader &b, int n, uint8_t s4[6][16], uint8_t s8[2][64], const Fallback &fb) { // Table 7-2
    for (int i = 0; i < n; i++) {
        bool present = b.u(1);
        if (i < 6) {
            if (present) {
                if (scaling_list(b, s4[i], 16)) memcpy(s4[i], i < 3 ? mi_default4x4_intra : mi_default4x4_inter, 16);
            } else if (i == 0 || i == 3)
                memcpy(s4[i], i == 0 ? fb.i4 : fb.p4, 16);
            else
                memcpy(s4[i], s4[i - 1], 16);
        } else if (i < 8) {
            int k = i - 6;
            if (present) {
                if (scaling_list(b, s8[k], 64)) memcpy(s8[k], k ? mi_default8x8_inter : mi_default8x8_intra, 64);
            } else
                memcpy(s8[k], k ? fb.p8 : fb.i8, 64);
        } else if (present) { // 4:4:4 chroma 8x8 lists: parsed, unused
            uint8_t tmp[64];
            scaling_list(b, tmp, 64);
        }
    }
}

static void hrd(BitReader &b, h264mi_sps *s) { // E.1.2 (A13: lengths after the loop)
    s->cpb_cnt_minus1 = b.ue();
    s->bit_rate_scale = b.u(4);
    s->cpb_size_scale = b.u(4);
    for (int i = 0; i <= s->cpb_cnt_minus1 && i < 32; i++) {
        b.ue();
        b.ue();
        b.u(1);
    }
    s->initial_cpb_removal_delay_length_minus1 = b.u(5);
    s->cpb_removal_delay_length_minus1 = b.u(5);
    s->dpb_output_delay_length_minus1 = b.u(5);
    s->time_offset_length = b.u(5);
}

// NewSPS (h264/sps.go:192-437)
int parse_sps(const uint8_t *rbsp, size_t len, h264mi_sps *s) {
    if (!rbsp || !s) return H264MI_EINVAL;
    BitReader b(rbsp, len);
    memset(s, 0, sizeof(*s));
    s->profile = b.u(8);
    s->constraint_flags = b.u(8);
    s->level = b.u(8);
    s->id = b.ue();
    s->chroma_format = 1;
    memset(s->scaling_list_4x4, 16, sizeof(s->scaling_list_4x4));
    memset(s->scaling_list_8x8, 16, sizeof(s->scaling_list_8x8));
    switch (s->profile) { // A11: only these profiles carry chroma_format_idc (list as h264/sps.go:230)
    case 100: case 110: case 122: case 244: case 44: case 83: case 86: case 118: case 128: case 138: case 139: case 134: case 135:
        s->chroma_format = b.ue();
        if (s->chroma_format == 3) s->use_separate_color_plane = b.u(1);
        s->bit_depth_luma_minus8 = b.ue();
        s->bit_depth_chroma_minus8 = b.ue();
        s->qprime_y_zero_transform_bypass = b.u(1);
        s->seq_scaling_matrix_present = b.u(1);
        if (s->seq_scaling_matrix_present) {
            Fallback fb{mi_default4x4_intra, mi_default4x4_inter, mi_default8x8_intra, mi_default8x8_inter};
            scaling_matrix(b, s->chroma_format != 3 ? 8 : 12, s->scaling_list_4x4, s->scaling_list_8x8, fb);
        }
        break;
    default: break;
    }
    s->log2_max_frame_num_minus4 = b.ue();
    s->pic_order_count_type = b.ue();
    if (s->pic_order_count_type == 0)
        s->log2_max_pic_order_cnt_lsb_min4 = b.ue();
    else if (s->pic_order_count_type == 1) {
        s->delta_pic_order_always_zero = b.u(1);
        s->offset_for_non_ref_pic = b.se();
        s->offset_for_top_to_bottom_field = b.se();
        s->num_ref_frames_in_pic_order_cnt_cycle = b.ue();
        if (s->num_ref_frames_in_pic_order_cnt_cycle > 255) return H264MI_EBITSTREAM;
        for (int i = 0; i < s->num_ref_frames_in_pic_order_cnt_cycle; i++) s->offset_for_ref_frame_list[i] = b.se();
    }
    s->max_num_ref_frames = b.ue();
    s->gaps_in_frame_num_value_allowed = b.u(1);
    s->pic_width_in_mbs_minus1 = b.ue();
    s->pic_height_in_map_units_minus1 = b.ue();
    s->frame_mbs_only = b.u(1);
    if (!s->frame_mbs_only) s->mb_adaptive_frame_field = b.u(1);
    s->direct_8x8_inference = b.u(1);
    s->frame_cropping = b.u(1);
    if (s->frame_cropping) {
        s->frame_crop_left_offset = b.ue();
        s->frame_crop_right_offset = b.ue();
        s->frame_crop_top_offset = b.ue();
        s->frame_crop_bottom_offset = b.ue();
    }
    s->vui_parameters_present = b.u(1);
    if (s->vui_parameters_present) {
        s->aspect_ratio_info_present = b.u(1);
        if (s->aspect_ratio_info_present) {
            s->aspect_ratio = b.u(8);
            if (s->aspect_ratio == 255) { // Extended_SAR (A12)
                s->sar_width = b.u(16);
                s->sar_height = b.u(16);
            }
        }
        s->overscan_info_present = b.u(1);
        if (s->overscan_info_present) s->overscan_appropriate = b.u(1);
        s->video_signal_type_present = b.u(1);
        if (s->video_signal_type_present) {
            s->video_format = b.u(3);
            s->video_full_range = b.u(1);
            s->color_description_present = b.u(1);
            if (s->color_description_present) {
                s->color_primaries = b.u(8);
                s->transfer_characteristics = b.u(8);
                s->matrix_coefficients = b.u(8);
            }
        }
        s->chroma_loc_info_present = b.u(1);
        if (s->chroma_loc_info_present) {
            s->chroma_sample_loc_type_top_field = b.ue();
            s->chroma_sample_loc_type_bottom_field = b.ue();
        }
        s->timing_info_present = b.u(1);
        if (s->timing_info_present) {
            s->num_units_in_tick = b.u(32);
            s->time_scale = b.u(32);
            s->fixed_frame_rate = b.u(1);
        }
        s->nal_hrd_parameters_present = b.u(1);
        if (s->nal_hrd_parameters_present) hrd(b, s);
        s->vcl_hrd_parameters_present = b.u(1);
        if (s->vcl_hrd_parameters_present) hrd(b, s);
        if (s->nal_hrd_parameters_present || s->vcl_hrd_parameters_present) s->low_hrd_delay = b.u(1);
        s->pic_struct_present = b.u(1);
        s->bitstream_restriction = b.u(1);
        if (s->bitstream_restriction) {
            s->motion_vectors_over_pic_boundaries = b.u(1);
            s->max_bytes_per_pic_denom = b.ue();
            s->max_bits_per_mb_denom = b.ue();
            s->log2_max_mv_length_horizontal = b.ue();
            s->log2_max_mv_length_vertical = b.ue();
            s->max_num_reorder_frames = b.ue();
            s->max_dec_frame_buffering = b.ue();
        }
    }
    if (b.overrun() || s->id > 31) {
        set_error("SPS: truncated or bad id");
        return H264MI_EBITSTREAM;
    }
    // Range checks (7.4.2.1.1, Annex A): these fields later size buffers, shift counts and addresses, and the input may
    // come straight from a socket -- ue(v) can carry anything up to 2^32 - 2.
    {
        const uint32_t wmb1 = static_cast<uint32_t>(s->pic_width_in_mbs_minus1), hmu1 = static_cast<uint32_t>(s->pic_height_in_map_units_minus1);
        const char *bad = nullptr;
        if (static_cast<uint32_t>(s->log2_max_frame_num_minus4) > 12) bad = "log2_max_frame_num_minus4 > 12";
        else if (static_cast<uint32_t>(s->pic_order_count_type) > 2) bad = "pic_order_cnt_type > 2";
        else if (static_cast<uint32_t>(s->log2_max_pic_order_cnt_lsb_min4) > 12) bad = "log2_max_pic_order_cnt_lsb_minus4 > 12";
        else if (static_cast<uint32_t>(s->max_num_ref_frames) > 16) bad = "max_num_ref_frames > 16";
        else if (wmb1 >= 512) bad = "pic_width_in_mbs > 512";
        else if (hmu1 >= 320 || (hmu1 + 1) * (2 - s->frame_mbs_only) > 320) bad = "pic_height_in_mbs > 320";
        else if (static_cast<uint32_t>(s->chroma_format) > 3) bad = "chroma_format_idc > 3";
        else if (static_cast<uint32_t>(s->bit_depth_luma_minus8) > 6 || static_cast<uint32_t>(s->bit_depth_chroma_minus8) > 6) bad = "bit depth > 14";
        else {
            // cropping (7-18 .. 7-21) in 4:2:0 frame units of 2 luma samples: what remains must be a non-empty part of the coded picture
            const uint32_t cl = static_cast<uint32_t>(s->frame_crop_left_offset), cr = static_cast<uint32_t>(s->frame_crop_right_offset);
            const uint32_t ct = static_cast<uint32_t>(s->frame_crop_top_offset), cb = static_cast<uint32_t>(s->frame_crop_bottom_offset);
            const uint32_t W = (wmb1 + 1) * 16, H = (hmu1 + 1) * (2 - s->frame_mbs_only) * 16, vy = 2 * (2 - s->frame_mbs_only);
            if (cl > W || cr > W || 2 * (static_cast<uint64_t>(cl) + cr) >= W) bad = "horizontal cropping leaves no picture";
            else if (ct > H || cb > H || vy * (static_cast<uint64_t>(ct) + cb) >= H) bad = "vertical cropping leaves no picture";
        }
        if (bad) {
            set_error("SPS: %s", bad);
            return H264MI_EBITSTREAM;
        }
    }
    // PicWidthInMbs / PicHeightInMbs (h264/slice.go:159-176), cropped size (7-18..7-21, 4:2:0 frame)
    s->pic_width_in_mbs = s->pic_width_in_mbs_minus1 + 1;
    s->pic_height_in_mbs = (s->pic_height_in_map_units_minus1 + 1) * (2 - s->frame_mbs_only);
    s->width = s->pic_width_in_mbs * 16 - 2 * (s->frame_crop_left_offset + s->frame_crop_right_offset);
    s->height = s->pic_height_in_mbs * 16 - 2 * (2 - s->frame_mbs_only) * (s->frame_crop_top_offset + s->frame_crop_bottom_offset);
    return H264MI_OK;
}

// NewPPS (h264/pps.go:40-133)
static int ceil_log2(uint32_t v) { // Ceil(Log2(v)), v >= 1
    int n = 0;
    while ((1u << n) < v) n++;
    return n;
}

int parse_pps(const h264mi_sps *sps, const uint8_t *rbsp, size_t len, h264mi_pps *p) { return parse_pps_ids(sps, rbsp, len, p, nullptr, 0, nullptr); }

// `ids` (optional, `cap` entries): slice_group_id[] of slice_group_map_type 6 -- too long for the POD (one byte per map unit)
int parse_pps_ids(const h264mi_sps *sps, const uint8_t *rbsp, size_t len, h264mi_pps *p, uint8_t *ids, size_t cap, size_t *n_ids) {
    if (!rbsp || !p || !sps) return H264MI_EINVAL;
    BitReader b(rbsp, len);
    memset(p, 0, sizeof(*p));
    if (n_ids) *n_ids = 0;
    // every ue(v) is range-checked as the unsigned value it is BEFORE it lands in an int32 field: a 32-bit Exp-Golomb code (31 leading
    // zeros) would otherwise turn negative and slip through signed comparisons (ids index tables, run lengths step loops)
    bool bad = false;
    auto UE = [&](uint32_t max) -> int32_t {
        const uint32_t v = b.ue();
        if (v > max) bad = true;
        return v > max ? 0 : static_cast<int32_t>(v);
    };
    p->id = UE(255);
    p->sps_id = UE(31);
    if (bad) {
        set_error("PPS: pic_parameter_set_id / seq_parameter_set_id out of range");
        return H264MI_EBITSTREAM;
    }
    p->entropy_coding_mode = b.u(1);
    p->bottom_field_pic_order_in_frame_present = b.u(1);
    p->num_slice_groups_minus1 = UE(7);
    if (bad) { // A.2: at most 8 slice groups in any profile
        set_error("PPS: num_slice_groups_minus1 out of range");
        return H264MI_EBITSTREAM;
    }
    if (p->num_slice_groups_minus1 > 0) { // 7.3.2.2 (h264/pps.go:57-80)
        const int ng = p->num_slice_groups_minus1 + 1;
        const uint32_t wmbs = static_cast<uint32_t>(sps->pic_width_in_mbs_minus1 + 1);
        const uint32_t map_units = wmbs * static_cast<uint32_t>(sps->pic_height_in_map_units_minus1 + 1);
        p->slice_group_map_type = UE(6);
        if (bad) {
            set_error("PPS: slice_group_map_type out of range");
            return H264MI_EBITSTREAM;
        }
        if (p->slice_group_map_type == 0) {
            for (int i = 0; i < ng; i++) p->run_length_minus1[i] = UE(map_units - 1); // 7.4.2.2: 0 .. PicSizeInMapUnits - 1
            if (bad) {
                set_error("PPS: run_length_minus1 beyond the picture");
                return H264MI_EBITSTREAM;
            }
        } else if (p->slice_group_map_type == 2) {
            for (int i = 0; i < ng - 1; i++) {
                const uint32_t tl = b.ue(), br = b.ue();
                // 7.4.2.2: top_left <= bottom_right < PicSizeInMapUnits, and its column not to the right of bottom_right's
                if (tl > br || br >= map_units || tl % wmbs > br % wmbs) {
                    set_error("PPS: slice group rectangle %d is malformed", i);
                    return H264MI_EBITSTREAM;
                }
                p->top_left[i] = static_cast<int32_t>(tl), p->bottom_right[i] = static_cast<int32_t>(br);
            }
        } else if (p->slice_group_map_type >= 3 && p->slice_group_map_type <= 5) {
            p->slice_group_change_direction = b.u(1);
            p->slice_group_change_rate_minus1 = UE(map_units - 1);
            if (ng != 2 || bad) {
                set_error("PPS: slice_group_map_type %d needs two slice groups and a change rate below the picture size", p->slice_group_map_type);
                return H264MI_EBITSTREAM;
            }
        } else if (p->slice_group_map_type == 6) {
            const uint32_t psm1 = b.ue();
            if (psm1 != map_units - 1) {
                set_error("PPS: pic_size_in_map_units_minus1 %u does not match the SPS (%u map units)", psm1, map_units);
                return H264MI_EBITSTREAM;
            }
            p->pic_size_in_map_units_minus1 = static_cast<int32_t>(psm1);
            const int bits = ceil_log2(static_cast<uint32_t>(ng));
            for (uint32_t i = 0; i < map_units && !b.overrun(); i++) {
                const uint32_t v = b.u(bits);
                if (v >= static_cast<uint32_t>(ng)) {
                    set_error("PPS: slice_group_id[%u] = %u out of range", i, v);
                    return H264MI_EBITSTREAM;
                }
                if (ids && i < cap) ids[i] = static_cast<uint8_t>(v);
            }
            if (n_ids) *n_ids = map_units;
        }
    }
    p->num_ref_idx_l0_default_active_minus1 = UE(31);
    p->num_ref_idx_l1_default_active_minus1 = UE(31);
    if (bad) {
        set_error("PPS: num_ref_idx_default_active_minus1 out of range");
        return H264MI_EBITSTREAM;
    }
    p->weighted_pred = b.u(1);
    p->weighted_bipred = b.u(2);
    p->pic_init_qp_minus26 = b.se();
    p->pic_init_qs_minus26 = b.se();
    p->chroma_qp_index_offset = b.se();
    p->deblocking_filter_control_present = b.u(1);
    p->constrained_intra_pred = b.u(1);
    p->redundant_pic_cnt_present = b.u(1);
    p->second_chroma_qp_index_offset = p->chroma_qp_index_offset;
    memcpy(p->scaling_list_4x4, sps->scaling_list_4x4, sizeof(p->scaling_list_4x4));
    memcpy(p->scaling_list_8x8, sps->scaling_list_8x8, sizeof(p->scaling_list_8x8));
    if (b.more_rbsp_data()) { // A15/A16
        p->transform_8x8_mode = b.u(1);
        p->pic_scaling_matrix_present = b.u(1);
        if (p->pic_scaling_matrix_present) {
            int n = 6 + (sps->chroma_format != 3 ? 2 : 6) * p->transform_8x8_mode;
            Fallback fb = sps->seq_scaling_matrix_present
                              ? Fallback{sps->scaling_list_4x4[0], sps->scaling_list_4x4[3], sps->scaling_list_8x8[0], sps->scaling_list_8x8[1]}
                              : Fallback{mi_default4x4_intra, mi_default4x4_inter, mi_default8x8_intra, mi_default8x8_inter};
            scaling_matrix(b, n, p->scaling_list_4x4, p->scaling_list_8x8, fb);
        }
        p->second_chroma_qp_index_offset = b.se();
    }
    // (pic_init_qp_minus26: -(26 + QpBdOffsetY) .. 25 by 7.4.2.2; encoders that lean on slice_qp_delta to come back into range exist -- the bound here only
    // keeps SliceQPY = 26 + pic_init_qp_minus26 + slice_qp_delta, which IS checked (0..51), free of overflow)
    if (b.overrun() || p->pic_init_qp_minus26 < -128 || p->pic_init_qp_minus26 > 127 || p->chroma_qp_index_offset < -12 || p->chroma_qp_index_offset > 12 ||
        p->second_chroma_qp_index_offset < -12 || p->second_chroma_qp_index_offset > 12) {
        set_error("PPS: truncated, or a QP field out of range");
        return H264MI_EBITSTREAM;
    }
    return H264MI_OK;
}

// slice_header() 7.3.3 (h264/slice.go:857-1032; A18 frame_num, A19 override flag, A20 MMCO loop)
int parse_slice_header(const h264mi_sps *s, const h264mi_pps *p, int nal_ref_idc, int nal_unit_type, const uint8_t *rbsp, size_t len,
                       h264mi_slice_header *sh) {
    if (!s || !p || !rbsp || !sh) return H264MI_EINVAL;
    BitReader b(rbsp, len);
    memset(sh, 0, sizeof(*sh));
    const bool idr = nal_unit_type == 5;
    sh->nal_ref_idc = nal_ref_idc;
    sh->nal_unit_type = nal_unit_type;
    // (every ue(v) is checked as an unsigned value before it is stored in an int32 field: see parse_pps_ids)
    bool bad = false;
    auto UE = [&](uint32_t max) -> int32_t {
        const uint32_t v = b.ue();
        if (v > max) bad = true;
        return v > max ? 0 : static_cast<int32_t>(v);
    };
    // PicSizeInMbs of the largest picture the library accepts (8192 x 5120 samples): the caller compares with the picture's own size
    sh->first_mb_in_slice = UE(512 * 320 - 1);
    sh->slice_type = UE(9);
    if (bad) return H264MI_EBITSTREAM;
    const int st = sh->slice_type % 5;
    sh->pps_id = UE(255);
    if (s->use_separate_color_plane) sh->color_plane_id = b.u(2);
    sh->frame_num = b.u(s->log2_max_frame_num_minus4 + 4);
    if (!s->frame_mbs_only) {
        sh->field_pic = b.u(1);
        if (sh->field_pic) sh->bottom_field = b.u(1);
    }
    if (idr) sh->idr_pic_id = UE(65535);
    if (s->pic_order_count_type == 0) {
        sh->pic_order_cnt_lsb = b.u(s->log2_max_pic_order_cnt_lsb_min4 + 4);
        if (p->bottom_field_pic_order_in_frame_present && !sh->field_pic) sh->delta_pic_order_cnt_bottom = b.se();
    }
    if (s->pic_order_count_type == 1 && !s->delta_pic_order_always_zero) {
        sh->delta_pic_order_cnt[0] = b.se();
        if (p->bottom_field_pic_order_in_frame_present && !sh->field_pic) sh->delta_pic_order_cnt[1] = b.se();
    }
    if (p->redundant_pic_cnt_present) sh->redundant_pic_cnt = UE(127);
    if (bad) return H264MI_EBITSTREAM;
    if (st == 1) sh->direct_spatial_mv_pred = b.u(1);
    sh->num_ref_idx_l0_active_minus1 = p->num_ref_idx_l0_default_active_minus1;
    sh->num_ref_idx_l1_active_minus1 = p->num_ref_idx_l1_default_active_minus1;
    if (st == 0 || st == 3 || st == 1) {
        sh->num_ref_idx_active_override = b.u(1);
        if (sh->num_ref_idx_active_override) {
            sh->num_ref_idx_l0_active_minus1 = UE(31);
            if (st == 1) sh->num_ref_idx_l1_active_minus1 = UE(31);
        }
        if (bad || sh->num_ref_idx_l0_active_minus1 > 31 || sh->num_ref_idx_l1_active_minus1 > 31) return H264MI_EBITSTREAM;
    }
    if (st != 2 && st != 4) { // ref_pic_list_modification(): list 0, then list 1 of B slices
        for (int l = 0; l < (st == 1 ? 2 : 1); l++) {
            int32_t &flag = l ? sh->ref_pic_list_modification_flag_l1 : sh->ref_pic_list_modification_flag_l0;
            int32_t &n = l ? sh->n_ref_pic_list_modifications_l1 : sh->n_ref_pic_list_modifications;
            int32_t *idcs = l ? sh->modification_of_pic_nums_l1 : sh->modification_of_pic_nums, *vals = l ? sh->modification_value_l1 : sh->modification_value;
            flag = b.u(1);
            if (flag)
                for (;;) {
                    uint32_t idc = b.ue();
                    if (idc == 3) break;
                    if (idc > 3 || n >= 66 || b.overrun()) return H264MI_EBITSTREAM;
                    idcs[n] = idc;
                    vals[n++] = UE((2u << 16) - 1); // abs_diff_pic_num_minus1 < MaxPicNum <= 2^17; long_term_pic_num < 32
                    if (bad) return H264MI_EBITSTREAM;
                }
        }
    }
    if ((p->weighted_pred && (st == 0 || st == 3)) || (p->weighted_bipred == 1 && st == 1)) { // pred_weight_table()
        const bool chroma = s->chroma_format != 0; // ChromaArrayType != 0: chroma_log2_weight_denom and the chroma weights are there (monochrome: h264/sps.go:226-243)
        sh->luma_log2_weight_denom = UE(7);
        sh->chroma_log2_weight_denom = chroma ? UE(7) : 0;
        if (bad) return H264MI_EBITSTREAM;
        for (int l = 0; l < (st == 1 ? 2 : 1); l++) {
            const int n = l ? sh->num_ref_idx_l1_active_minus1 : sh->num_ref_idx_l0_active_minus1;
            int32_t *lf = l ? sh->luma_weight_l1_flag : sh->luma_weight_l0_flag, *lw = l ? sh->luma_weight_l1 : sh->luma_weight_l0;
            int32_t *lo = l ? sh->luma_offset_l1 : sh->luma_offset_l0, *cf = l ? sh->chroma_weight_l1_flag : sh->chroma_weight_l0_flag;
            int32_t(*cw)[2] = l ? sh->chroma_weight_l1 : sh->chroma_weight_l0, (*co)[2] = l ? sh->chroma_offset_l1 : sh->chroma_offset_l0;
            for (int i = 0; i <= n; i++) {
                lw[i] = 1 << sh->luma_log2_weight_denom;
                cw[i][0] = cw[i][1] = 1 << sh->chroma_log2_weight_denom;
                lf[i] = b.u(1);
                // 7.4.3.2: CODED weights and offsets are in -128..127; the inferred default 2^denom (128 for denom 7) is not coded
                if (lf[i]) {
                    lw[i] = b.se();
                    lo[i] = b.se();
                    if (lw[i] < -128 || lw[i] > 127 || lo[i] < -128 || lo[i] > 127) return H264MI_EBITSTREAM;
                }
                cf[i] = chroma ? b.u(1) : 0;
                if (cf[i])
                    for (int j = 0; j < 2; j++) {
                        cw[i][j] = b.se();
                        co[i][j] = b.se();
                        if (cw[i][j] < -128 || cw[i][j] > 127 || co[i][j] < -128 || co[i][j] > 127) return H264MI_EBITSTREAM;
                    }
                if (b.overrun()) return H264MI_EBITSTREAM;
            }
        }
    }
    if (nal_ref_idc != 0) { // dec_ref_pic_marking()
        if (idr) {
            sh->no_output_of_prior_pics_flag = b.u(1);
            sh->long_term_reference_flag = b.u(1);
        } else {
            sh->adaptive_ref_pic_marking_mode_flag = b.u(1);
            if (sh->adaptive_ref_pic_marking_mode_flag)
                for (;;) {
                    uint32_t op = b.ue();
                    if (op == 0) break;
                    int k = sh->n_memory_management_control_operations;
                    if (op > 6 || k >= 66 || b.overrun()) return H264MI_EBITSTREAM;
                    sh->memory_management_control_operation[k] = op;
                    if (op == 1 || op == 3 || op == 2 || op == 4) sh->mmco_arg1[k] = UE((2u << 16) - 1);
                    if (op == 3 || op == 6) sh->mmco_arg2[k] = UE(32);
                    if (bad) return H264MI_EBITSTREAM;
                    sh->n_memory_management_control_operations++;
                }
        }
    }
    if (p->entropy_coding_mode && st != 2 && st != 4) {
        sh->cabac_init = UE(2);
        if (bad) return H264MI_EBITSTREAM;
    }
    sh->slice_qp_delta = b.se();
    if (sh->slice_qp_delta < -256 || sh->slice_qp_delta > 256) return H264MI_EBITSTREAM; // (SliceQPY itself is checked below: this keeps the sum free of overflow)
    if (st == 3 || st == 4) {
        if (st == 3) sh->sp_for_switch = b.u(1);
        sh->slice_qs_delta = b.se();
    }
    if (p->deblocking_filter_control_present) {
        sh->disable_deblocking_filter = UE(2);
        if (bad) return H264MI_EBITSTREAM;
        if (sh->disable_deblocking_filter != 1) {
            sh->slice_alpha_c0_offset_div2 = b.se();
            sh->slice_beta_offset_div2 = b.se();
        }
        if (sh->disable_deblocking_filter > 2 || sh->slice_alpha_c0_offset_div2 < -6 || sh->slice_alpha_c0_offset_div2 > 6 ||
            sh->slice_beta_offset_div2 < -6 || sh->slice_beta_offset_div2 > 6)
            return H264MI_EBITSTREAM;
    }
    if (p->num_slice_groups_minus1 > 0 && p->slice_group_map_type >= 3 && p->slice_group_map_type <= 5) {
        // Ceil(Log2(PicSizeInMapUnits / SliceGroupChangeRate + 1)) bits, "/" exact (7.4.3; h264/slice.go:1028-1031 divides the
        // minus1 values as integers): the smallest n with (2^n - 1) * rate >= map units
        const uint64_t units = static_cast<uint64_t>(s->pic_width_in_mbs_minus1 + 1) * static_cast<uint64_t>(s->pic_height_in_map_units_minus1 + 1);
        const uint64_t rate = static_cast<uint64_t>(p->slice_group_change_rate_minus1) + 1;
        int n = 0;
        while (((1ull << n) - 1) * rate < units) n++;
        sh->slice_group_change_cycle = n ? static_cast<int32_t>(b.u(n)) : 0;
        if (static_cast<uint64_t>(sh->slice_group_change_cycle) > (units + rate - 1) / rate) return H264MI_EBITSTREAM; // 7.4.3 range
    }
    sh->slice_qp_y = 26 + p->pic_init_qp_minus26 + sh->slice_qp_delta; // (7-30), h264/cabac.go:113
    sh->slice_data_bit_offset = b.pos();
    if (b.overrun() || sh->slice_qp_y < 0 || sh->slice_qp_y > 51) {
        set_error("slice header: truncated or QP out of range");
        return H264MI_EBITSTREAM;
    }
    return H264MI_OK;
}

// ---------------------------------------------------------------- slice groups (FMO), 8.2.2
// MapUnitToSliceGroupMap (h264/slice.go:457-529; types 3-6 are TODO there, type 0 / type 2 index past their arrays).
int map_unit_to_slice_group_map(const h264mi_sps *s, const h264mi_pps *p, const uint8_t *ids, size_t n_ids, int cycle, uint8_t *map, size_t cap, size_t *n_out) {
    if (!s || !p || !map) return H264MI_EINVAL;
    const int W = s->pic_width_in_mbs_minus1 + 1, Hm = s->pic_height_in_map_units_minus1 + 1;
    const size_t units = static_cast<size_t>(W) * Hm;
    if (n_out) *n_out = units;
    if (cap < units) return H264MI_ECAPACITY;
    if (W < 1 || Hm < 1 || W > 512 || Hm > 320) return H264MI_EINVAL;
    const int ng = p->num_slice_groups_minus1 + 1;
    if (ng < 1 || ng > 8) return H264MI_EBITSTREAM;
    if (ng == 1) {
        memset(map, 0, units);
        return H264MI_OK;
    }
    // a public entry point with a caller-supplied PPS (h264mi_map_unit_to_slice_group_map): the fields the loops below step by or
    // index with are checked again here, as unsigned values (7.4.2.2)
    if (p->slice_group_map_type == 0) {
        for (int g = 0; g < ng; g++)
            if (static_cast<uint32_t>(p->run_length_minus1[g]) >= units) return H264MI_EBITSTREAM;
    } else if (p->slice_group_map_type == 2) {
        for (int g = 0; g < ng - 1; g++) {
            const uint32_t tl = static_cast<uint32_t>(p->top_left[g]), br = static_cast<uint32_t>(p->bottom_right[g]);
            if (tl > br || br >= units || tl % W > br % W) return H264MI_EBITSTREAM;
        }
    } else if (p->slice_group_map_type >= 3 && p->slice_group_map_type <= 5) {
        if (ng != 2 || static_cast<uint32_t>(p->slice_group_change_rate_minus1) >= units) return H264MI_EBITSTREAM;
    }
    const int flag = p->slice_group_change_direction ? 1 : 0;
    const size_t rate = static_cast<size_t>(p->slice_group_change_rate_minus1) + 1;
    const size_t in_group0 = std::min(static_cast<size_t>(cycle > 0 ? cycle : 0) * rate, units); // MapUnitsInSliceGroup0 (7-33)
    const size_t upper_left = flag ? units - in_group0 : in_group0;                              // sizeOfUpperLeftGroup (8-14)
    switch (p->slice_group_map_type) {
    case 0: { // 8.2.2.1 interleaved
        size_t i = 0;
        do {
            for (int g = 0; g < ng && i < units; i += static_cast<size_t>(p->run_length_minus1[g++]) + 1)
                for (size_t j = 0; j <= static_cast<size_t>(p->run_length_minus1[g]) && i + j < units; j++) map[i + j] = static_cast<uint8_t>(g);
        } while (i < units);
        break;
    }
    case 1: // 8.2.2.2 dispersed
        for (size_t i = 0; i < units; i++) map[i] = static_cast<uint8_t>(((i % W) + (((i / W) * ng) / 2)) % ng);
        break;
    case 2: // 8.2.2.3 foreground rectangles and a left-over group
        memset(map, ng - 1, units);
        for (int g = ng - 2; g >= 0; g--) {
            const int y0 = p->top_left[g] / W, x0 = p->top_left[g] % W, y1 = p->bottom_right[g] / W, x1 = p->bottom_right[g] % W;
            for (int y = y0; y <= y1 && y < Hm; y++)
                for (int x = x0; x <= x1; x++) map[static_cast<size_t>(y) * W + x] = static_cast<uint8_t>(g);
        }
        break;
    case 3: { // 8.2.2.4 box-out
        memset(map, 1, units);
        int x = (W - flag) / 2, y = (Hm - flag) / 2;
        int left = x, top = y, right = x, bottom = y, xd = flag - 1, yd = flag;
        for (size_t k = 0; k < in_group0;) {
            uint8_t &m = map[static_cast<size_t>(y) * W + x];
            const bool vacant = m == 1;
            if (vacant) m = 0, k++;
            if (xd == -1 && x == left) {
                left = std::max(left - 1, 0), x = left, xd = 0, yd = 2 * flag - 1;
            } else if (xd == 1 && x == right) {
                right = std::min(right + 1, W - 1), x = right, xd = 0, yd = 1 - 2 * flag;
            } else if (yd == -1 && y == top) {
                top = std::max(top - 1, 0), y = top, xd = 1 - 2 * flag, yd = 0;
            } else if (yd == 1 && y == bottom) {
                bottom = std::min(bottom + 1, Hm - 1), y = bottom, xd = 2 * flag - 1, yd = 0;
            } else
                x += xd, y += yd;
        }
        break;
    }
    case 4: // 8.2.2.5 raster scan
        for (size_t i = 0; i < units; i++) map[i] = static_cast<uint8_t>(i < upper_left ? flag : 1 - flag);
        break;
    case 5: { // 8.2.2.6 wipe
        size_t k = 0;
        for (int j = 0; j < W; j++)
            for (int i = 0; i < Hm; i++) map[static_cast<size_t>(i) * W + j] = static_cast<uint8_t>(k++ < upper_left ? flag : 1 - flag);
        break;
    }
    case 6: // 8.2.2.7 explicit
        if (!ids || n_ids < units) {
            set_error("slice_group_map_type 6 needs the slice_group_id array of the PPS (h264mi_pps_slice_group_ids)");
            return H264MI_EINVAL;
        }
        memcpy(map, ids, units);
        break;
    default:
        return H264MI_EBITSTREAM;
    }
    return H264MI_OK;
}

// MbToSliceGroupMap 8.2.2.8 (h264/slice.go:134-158) for the pictures this library decodes: frames of frame_mbs_only streams
// and frame pictures of non-MBAFF interlace streams (a map unit = two macroblock rows); field_pic: a field picture of such a stream.
int mb_to_slice_group_map(const h264mi_sps *s, const h264mi_pps *p, const uint8_t *ids, size_t n_ids, int cycle, int field_pic, uint8_t *map, size_t cap, size_t *n_out) {
    if (!s || !p || !map) return H264MI_EINVAL;
    const int W = s->pic_width_in_mbs_minus1 + 1, Hm = s->pic_height_in_map_units_minus1 + 1;
    const size_t units = static_cast<size_t>(W) * Hm;
    const bool mbaff = s->mb_adaptive_frame_field && !field_pic;
    const bool direct = s->frame_mbs_only || field_pic; // one map unit per macroblock
    const size_t n_mbs = direct ? units : 2 * units;
    if (n_out) *n_out = n_mbs;
    if (cap < n_mbs) return H264MI_ECAPACITY;
    if (direct) return map_unit_to_slice_group_map(s, p, ids, n_ids, cycle, map, cap, nullptr);
    std::vector<uint8_t> mu(units);
    const int r = map_unit_to_slice_group_map(s, p, ids, n_ids, cycle, mu.data(), units, nullptr);
    if (r != H264MI_OK) return r;
    for (size_t i = 0; i < n_mbs; i++) map[i] = mbaff ? mu[i / 2] : mu[(i / (2 * static_cast<size_t>(W))) * W + (i % W)];
    return H264MI_OK;
}

// nextMbAddress (8-17; h264/slice.go:530-552 compares an entry with itself): the next macroblock of n's slice group, or n_mbs
int next_mb_address(const uint8_t *map, size_t n_mbs, size_t n) {
    if (!map || n >= n_mbs) return static_cast<int>(n_mbs);
    size_t i = n + 1;
    while (i < n_mbs && map[i] != map[n]) i++;
    return static_cast<int>(i);
}

} // namespace mi

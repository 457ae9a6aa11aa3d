/*
 * include/h264mi.h -- C ABI of libh264mi.so: MI355X-native H.264 Annex-B decode path.
 *
 * This is the drop-in boundary for the hot path of mrmod/h264decode's Go package `h264`
 * (Annex-B bytes -> NAL -> SPS/PPS/slice header -> macroblocks -> Y/Cb/Cr planes).  The reference
 * has no FFI of its own ("No interface contracts are implemented right now", README.md:4): the
 * boundary is the set of exported Go identifiers; each entry point below names the reference
 * function(s) it replaces.  The cgo / ctypes bindings are shown in INTEGRATION.md.
 *
 * Rules:
 *  - extern "C", plain pointers and sizes only; no C++/torch types.
 *  - every function returns an int32 status (0 = OK, negative = H264MI_E*); nothing aborts or
 *    throws across the boundary (the reference panics/os.Exit()s: h264/server.go:136-143).
 *  - caller owns input buffers for the duration of the call only; the library owns device memory.
 *  - threading: a decoder handle is used by one thread at a time (any thread: every entry point selects the decoder's
 *    HIP device for the duration of the call and restores the caller's); distinct handles are independent.  The error
 *    text of h264mi_last_error_string() is thread-local: fetch it on the thread that received the status (a Go caller
 *    wraps call + fetch in runtime.LockOSThread).
 *  - the pixel path runs ONLY on the GPU (HIP, gfx950).  There is no CPU fallback: without a
 *    usable device h264mi_init / h264mi_decoder_create fail with H264MI_ENODEVICE.
 *
 * Field names follow the reference structs in snake_case (Go: CamelCase): NalUnit
 * h264/nalUnit.go:3-30, SPS h264/sps.go:9-103, PPS h264/pps.go:10-38, SliceHeader
 * h264/slice.go:23-75.  Values are spec-correct where the reference is not.
 */
#ifndef H264MI_H
#define H264MI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H264MI_OK 0
#define H264MI_EINVAL (-1)      /* bad argument */
#define H264MI_EBITSTREAM (-2)  /* malformed / truncated syntax */
#define H264MI_EUNSUPPORTED (-3)/* valid H.264 outside the implemented scope (MBAFF, CABAC-coded field pictures unless asked for, SP/SI slices, data partitioning,
                                 * 4:2:2 / 4:4:4, more than 8 bits ...): the message says what and why.  Of SVC / MVC / 3D-AVC streams the base layer / base view
                                 * is decoded; their extension NAL units (14, 15, 20, 21) are passed over */
#define H264MI_ENODEVICE (-4)   /* no usable HIP device / kernel image */
#define H264MI_ENOMEM (-5)
#define H264MI_EDEVICE (-6)     /* HIP runtime error */
#define H264MI_ECAPACITY (-7)   /* caller buffer or decoder configuration too small */
#define H264MI_EDECODE (-8)     /* a GPU entropy kernel reported a slice error */

/* ---- NAL unit: h264/nalUnit.go:3-30 (NalUnit), :75-131 (NewNalUnit) ---- */
typedef struct {
    int32_t num_bytes;          /* NumBytes: NAL size incl. header */
    int32_t forbidden_zero_bit; /* ForbiddenZeroBit */
    int32_t ref_idc;            /* RefIdc */
    int32_t type;               /* Type */
    int32_t header_bytes;       /* HeaderBytes (1; 4 for types 14/20/21) */
    int32_t svc_extension_flag, avc_3d_extension_flag; /* SvcExtensionFlag, Avc3dExtensionFlag (parsed, Annex G/H/J not decoded) */
    int64_t offset;             /* byte offset of the NAL header inside the scanned buffer */
} h264mi_nal;

/* replaces isStartSequence/readNalUnit (h264/server.go:28-39,64-111): finds every NAL of an
 * Annex-B buffer (3- and 4-byte start codes, trailing zeros stripped).  *n receives the count;
 * returns H264MI_ECAPACITY if more than `cap` NALs exist (first `cap` are still written). */
int32_t h264mi_annexb_scan(const uint8_t *buf, size_t len, h264mi_nal *out, int32_t cap, int32_t *n);
/* replaces NewNalUnit + (*NalUnit).RBSP() (h264/nalUnit.go:72,75-131): parses the header of the
 * NAL at nal_bytes[0..len) and writes the RBSP (emulation prevention removed) to rbsp_out
 * (capacity >= len).  *rbsp_len receives its size. */
int32_t h264mi_nal_parse(const uint8_t *nal_bytes, size_t len, h264mi_nal *nal, uint8_t *rbsp_out, size_t *rbsp_len);

/* ---- SPS: h264/sps.go:9-103, NewSPS :192-437 ---- */
typedef struct {
    int32_t profile, constraint_flags, level, id;                      /* Profile, Constraint0..5 (packed), Level, ID */
    int32_t chroma_format, use_separate_color_plane;                   /* ChromaFormat, UseSeparateColorPlane */
    int32_t bit_depth_luma_minus8, bit_depth_chroma_minus8;            /* BitDepthLumaMinus8, BitDepthChromaMinus8 */
    int32_t qprime_y_zero_transform_bypass, seq_scaling_matrix_present;/* QPrimeYZeroTransformBypass, SeqScalingMatrixPresent */
    int32_t log2_max_frame_num_minus4, pic_order_count_type, log2_max_pic_order_cnt_lsb_min4;
    int32_t delta_pic_order_always_zero, offset_for_non_ref_pic, offset_for_top_to_bottom_field;
    int32_t num_ref_frames_in_pic_order_cnt_cycle;
    int32_t offset_for_ref_frame_list[256];                            /* OffsetForRefFrameList */
    int32_t max_num_ref_frames, gaps_in_frame_num_value_allowed;
    int32_t pic_width_in_mbs_minus1, pic_height_in_map_units_minus1;
    int32_t frame_mbs_only, mb_adaptive_frame_field, direct_8x8_inference;
    int32_t frame_cropping, frame_crop_left_offset, frame_crop_right_offset, frame_crop_top_offset, frame_crop_bottom_offset;
    int32_t vui_parameters_present;
    int32_t aspect_ratio_info_present, aspect_ratio, sar_width, sar_height;
    int32_t overscan_info_present, overscan_appropriate;
    int32_t video_signal_type_present, video_format, video_full_range, color_description_present;
    int32_t color_primaries, transfer_characteristics, matrix_coefficients;
    int32_t chroma_loc_info_present, chroma_sample_loc_type_top_field, chroma_sample_loc_type_bottom_field;
    int32_t timing_info_present;
    uint32_t num_units_in_tick, time_scale;
    int32_t fixed_frame_rate;
    int32_t nal_hrd_parameters_present, vcl_hrd_parameters_present, low_hrd_delay, pic_struct_present;
    int32_t cpb_cnt_minus1, bit_rate_scale, cpb_size_scale;
    int32_t initial_cpb_removal_delay_length_minus1, cpb_removal_delay_length_minus1, dpb_output_delay_length_minus1, time_offset_length;
    int32_t bitstream_restriction, motion_vectors_over_pic_boundaries, max_bytes_per_pic_denom, max_bits_per_mb_denom;
    int32_t log2_max_mv_length_horizontal, log2_max_mv_length_vertical, max_num_reorder_frames, max_dec_frame_buffering;
    /* resolved scaling lists (Table 7-2 fall-back applied), zig-zag order */
    uint8_t scaling_list_4x4[6][16];
    uint8_t scaling_list_8x8[2][64];
    /* derived (h264/slice.go:159-176 PicWidthInMbs ... PicSizeInMbs) */
    int32_t pic_width_in_mbs, pic_height_in_mbs, width, height;       /* width/height = cropped display size */
} h264mi_sps;
/* replaces NewSPS(rbsp, showPacket) (h264/sps.go:192) */
int32_t h264mi_sps_parse(const uint8_t *rbsp, size_t len, h264mi_sps *sps);

/* ---- PPS: h264/pps.go:10-38, NewPPS :40-133 ---- */
typedef struct {
    int32_t id, sps_id, entropy_coding_mode, bottom_field_pic_order_in_frame_present, num_slice_groups_minus1;
    int32_t num_ref_idx_l0_default_active_minus1, num_ref_idx_l1_default_active_minus1;
    int32_t weighted_pred, weighted_bipred, pic_init_qp_minus26, pic_init_qs_minus26, chroma_qp_index_offset;
    int32_t deblocking_filter_control_present, constrained_intra_pred, redundant_pic_cnt_present;
    int32_t transform_8x8_mode, pic_scaling_matrix_present, second_chroma_qp_index_offset;
    uint8_t scaling_list_4x4[6][16];
    uint8_t scaling_list_8x8[2][64];
    /* slice groups (num_slice_groups_minus1 > 0; h264/pps.go:16-23, :57-80).  slice_group_id[] of map type 6 is one entry
     * per map unit and does not live in this struct: h264mi_pps_slice_group_ids() */
    int32_t slice_group_map_type, run_length_minus1[8], top_left[8], bottom_right[8];
    int32_t slice_group_change_direction, slice_group_change_rate_minus1, pic_size_in_map_units_minus1;
} h264mi_pps;
/* replaces NewPPS(sps, rbsp, showPacket) (h264/pps.go:40).  `sps` is the SPS the PPS refers to
 * (the reference passes "the last SPS": h264/server.go:155). */
int32_t h264mi_pps_parse(const h264mi_sps *sps, const uint8_t *rbsp, size_t len, h264mi_pps *pps);
/* PPS.SliceGroupId (h264/pps.go:23, :72-78): slice_group_id[i] of a PPS with slice_group_map_type 6, one byte per map unit.
 * *n receives pic_size_in_map_units_minus1 + 1 (0 for any other PPS); at most cap entries are written. */
int32_t h264mi_pps_slice_group_ids(const h264mi_sps *sps, const uint8_t *rbsp, size_t len, uint8_t *ids, size_t cap, size_t *n);

/* ---- slice header: h264/slice.go:23-75, NewSliceContext :835-1048 ---- */
typedef struct {
    int32_t first_mb_in_slice, slice_type, pps_id, color_plane_id, frame_num;
    int32_t field_pic, bottom_field, idr_pic_id, pic_order_cnt_lsb, delta_pic_order_cnt_bottom;
    int32_t delta_pic_order_cnt[2], redundant_pic_cnt, direct_spatial_mv_pred;
    int32_t num_ref_idx_active_override, num_ref_idx_l0_active_minus1, num_ref_idx_l1_active_minus1;
    int32_t ref_pic_list_modification_flag_l0, n_ref_pic_list_modifications;
    int32_t modification_of_pic_nums[66], modification_value[66]; /* idc / abs_diff_pic_num_minus1 | long_term_pic_num */
    int32_t luma_log2_weight_denom, chroma_log2_weight_denom;
    int32_t luma_weight_l0_flag[32], luma_weight_l0[32], luma_offset_l0[32];
    int32_t chroma_weight_l0_flag[32], chroma_weight_l0[32][2], chroma_offset_l0[32][2];
    /* list 1 of B slices (7.3.3.1, 7.3.3.2) */
    int32_t ref_pic_list_modification_flag_l1, n_ref_pic_list_modifications_l1;
    int32_t modification_of_pic_nums_l1[66], modification_value_l1[66];
    int32_t luma_weight_l1_flag[32], luma_weight_l1[32], luma_offset_l1[32];
    int32_t chroma_weight_l1_flag[32], chroma_weight_l1[32][2], chroma_offset_l1[32][2];
    int32_t no_output_of_prior_pics_flag, long_term_reference_flag, adaptive_ref_pic_marking_mode_flag;
    int32_t n_memory_management_control_operations;
    int32_t memory_management_control_operation[66], mmco_arg1[66], mmco_arg2[66];
    int32_t cabac_init, slice_qp_delta, sp_for_switch, slice_qs_delta;
    int32_t disable_deblocking_filter, slice_alpha_c0_offset_div2, slice_beta_offset_div2;
    int32_t slice_group_change_cycle; /* slice group map types 3..5 (h264/slice.go:1028-1031) */
    /* derived */
    int32_t nal_ref_idc, nal_unit_type, slice_qp_y; /* SliceQPy (h264/cabac.go:113) */
    int64_t slice_data_bit_offset;                  /* where slice_data() starts inside the RBSP */
} h264mi_slice_header;
/* replaces NewSliceContext's header part (h264/slice.go:857-1032) */
int32_t h264mi_slice_header_parse(const h264mi_sps *sps, const h264mi_pps *pps, int32_t nal_ref_idc, int32_t nal_unit_type,
                                  const uint8_t *rbsp, size_t len, h264mi_slice_header *sh);

/* First slice of a new picture?  7.4.1.2.4 on two slice headers of one stream (`prev`: the first slice of the current picture), plus what
 * 7.4.3 makes constant over the slices of a picture (slice_group_change_cycle, the marking script) -- memory management operation 5
 * resets frame_num and the picture order count, so the headers of the next picture may agree with it in everything 7.4.1.2.4 lists.
 * Returns 1 / 0 (negative: H264MI_EINVAL).  What h264mi_batch_prepare applies itself; exported for front-ends that cut a byte stream
 * into access units (the reference reads NAL by NAL and never needs it: h264/server.go:113-166). */
int32_t h264mi_slice_starts_picture(const h264mi_sps *sps, const h264mi_slice_header *prev, const h264mi_slice_header *cur);

/* ---- slice groups (FMO, 8.2.2): h264/slice.go:134-158, :457-552 ----
 * MapUnitToSliceGroupMap(sps, pps, header) (h264/slice.go:457): map types 0..6 (the reference stops at 2).  ids / n_ids: the
 * slice_group_id array of a type-6 PPS (NULL / 0 otherwise); slice_group_change_cycle: the slice header's field (types 3..5).
 * *n receives PicSizeInMapUnits; H264MI_ECAPACITY if cap is smaller. */
int32_t h264mi_map_unit_to_slice_group_map(const h264mi_sps *sps, const h264mi_pps *pps, const uint8_t *ids, size_t n_ids,
                                           int32_t slice_group_change_cycle, uint8_t *map, size_t cap, size_t *n);
/* MbToSliceGroupMap(sps, pps, header) (h264/slice.go:134): 8.2.2.8, one entry per macroblock of the picture. */
int32_t h264mi_mb_to_slice_group_map(const h264mi_sps *sps, const h264mi_pps *pps, const uint8_t *ids, size_t n_ids,
                                     int32_t slice_group_change_cycle, int32_t field_pic, uint8_t *map, size_t cap, size_t *n);
/* nextMbAddress(n, ...) (h264/slice.go:530): the next macroblock of n's slice group in `map`, n_mbs if there is none. */
int32_t h264mi_next_mb_address(const uint8_t *map, size_t n_mbs, size_t n);

/* ---- GPU decode: replaces NewSliceData / MbPred and the absent L7 reconstruction
 *      (h264/slice.go:570-830, :252-454; README.md:8-10 TODO items) ---- */
typedef struct h264mi_decoder h264mi_decoder;

typedef struct {
    /* sizeof(h264mi_config) as the CALLER was compiled: the struct grows at its end from version to version, and a field that lies beyond
     * struct_size is taken as 0 (its default) instead of being read from whatever follows a shorter struct.  0 is refused: zero-initialise
     * the struct and set this field (H264MI_CONFIG_INIT).  Adding this field in front was a ONE-TIME ABI break (round 4): binaries built against
     * the header without it do not work with this library; size-based compatibility starts with this version of the struct. */
    uint32_t struct_size;
    int32_t device;                /* HIP device ordinal */
    int32_t max_streams;           /* independent streams decoded side by side */
    int32_t max_width, max_height; /* display size upper bound (coded size is rounded up to 16) */
    int32_t max_frames_per_batch;  /* PICTURES per stream and per h264mi_decode_batch call: a frame picture is one, a frame coded as two field
                                    * pictures (h264/slice.go:867-872 field_pic_flag) is two */
    int32_t max_slices_per_frame;
    int64_t max_bitstream_bytes;   /* per batch, summed over streams */
    void *hip_stream;              /* hipStream_t to launch on; NULL = a private stream */
    /* Sizing knobs, 0 = default.  max_ref_frames: the largest max_num_ref_frames (h264/sps.go:61) the streams will carry; the frame
     * pool holds that many reference slots per stream (default 16, the limit of any level; a 1080p slot is 3.1 MB per stream) and
     * a stream that declares more is refused with H264MI_ECAPACITY.  coef_blocks_per_mb: residual pool size in 32-byte blocks per
     * macroblock (default 8 of at most 26; see h264mi_decoder_coef_pool). */
    int32_t max_ref_frames, coef_blocks_per_mb;
    /* b_pictures: 0 = what only B pictures need (list-1 vectors, the motion of reference pictures kept for direct prediction: 16 GB for 256
     * streams of 1080p) is allocated when the first B slice arrives -- I / P deployments never pay for it, and the FIRST B picture a decoder
     * sees must find its co-located picture in the same batch or the one before (otherwise that stream is refused until its next IDR picture);
     * 1 = allocated at create time, motion kept from the first picture on. */
    int32_t b_pictures;
    /* allow_unpinned_field_cabac: 1 = field pictures (h264/slice.go:867-872 field_pic_flag) coded with CABAC are decoded.  The default is a refusal
     * (H264MI_EUNSUPPORTED): the initialisation values of the contexts only field-coded blocks use (ctxIdx 277-398, 436-459; four sets) were entered
     * into this library's tables without the standard at hand and nothing pins them -- no third-party field-coded stream, no reference table on the
     * build machine.  With a wrong value a slice loses synchronisation and does not end on end_of_slice_flag at the picture's last macroblock: such
     * slices fail (H264MI_EDECODE, the stream waits for its next IDR picture) and are counted: h264mi_decoder_unpinned_failures. */
    int32_t allow_unpinned_field_cabac;
} h264mi_config;
#define H264MI_CONFIG_INIT {(uint32_t)sizeof(h264mi_config)} /* h264mi_config cfg = H264MI_CONFIG_INIT; then set the fields */

typedef struct {
    int32_t n_frames;         /* frames decoded in this batch */
    int32_t n_slices;
    int64_t n_macroblocks;
    int64_t bitstream_bytes;  /* RBSP bytes resident on the device */
    int32_t width, height, coded_width, coded_height; /* of stream 0 */
    double host_prepare_ms;   /* NAL scan + header parse + DPB bookkeeping + upload enqueue */
} h264mi_batch_info;

int32_t h264mi_init(int32_t device);
int32_t h264mi_decoder_create(const h264mi_config *cfg, h264mi_decoder **out);
int32_t h264mi_decoder_destroy(h264mi_decoder *dec);
int32_t h264mi_decoder_set_stream(h264mi_decoder *dec, void *hip_stream);
/* Forget all reference pictures of every stream (seek / new sequence). */
int32_t h264mi_decoder_reset(h264mi_decoder *dec);
/* Forget everything about ONE stream slot -- parameter sets, reference pictures, POC / frame_num history -- before the
 * slot is given to a new connection (the reference starts every connection from scratch: h264/server.go:113-125). */
int32_t h264mi_stream_reset(h264mi_decoder *dec, int32_t stream);
/* Error isolation for batches of unrelated streams (one connection each).  Off (default): the first stream error fails
 * h264mi_batch_prepare / h264mi_batch_sync.  On: a stream whose chunk cannot be parsed, or whose slices fail in the
 * entropy kernel, is dropped from the batch and marked; the calls return H264MI_OK and the other streams decode
 * normally.  A marked stream resumes at its next IDR picture. */
int32_t h264mi_decoder_set_isolation(h264mi_decoder *dec, int32_t on);
/* Status of a stream in the current batch: H264MI_OK or the H264MI_E* that took it out (valid after prepare; entropy
 * kernel failures appear after sync).  Pipelined callers (execute(k); prepare(k + 1); execute(k + 1); ... one sync for
 * several batches): h264mi_batch_sync looks at every batch executed since the last sync, and h264mi_batch_prepare at the
 * batch whose staging set it takes back, so a failure in batch k marks its stream (references dropped, nothing decoded
 * before its next IDR picture) before batch k + 2 is parsed at the latest; batch k + 1 of that stream, prepared before the
 * failure was known, is decoded from the damaged pictures.  The status VALUE is reset by every prepare: read it after
 * the sync that follows an execute if it matters. */
int32_t h264mi_stream_status(h264mi_decoder *dec, int32_t stream, int32_t *status);

/* Stage 1 (host + H2D): scan and parse each stream's Annex-B chunk (whole access units), run
 * picture management (POC 8.2.1, reference lists 8.2.4, marking 8.2.5), and make the RBSP bytes
 * and slice/picture descriptors resident in device memory.  bufs[i]/lens[i] = chunk of stream i
 * (NULL/0 = nothing for that stream). */
int32_t h264mi_batch_prepare(h264mi_decoder *dec, int32_t n_streams, const uint8_t *const *bufs, const size_t *lens, h264mi_batch_info *info);
/* Stage 2 (GPU only): entropy-decode every slice of the prepared batch (one slice per wavefront),
 * then reconstruct and deblock picture by picture.  Asynchronous on the decoder's stream.  May be
 * called repeatedly for the same prepared batch (benchmarks); results are identical each time. */
int32_t h264mi_batch_execute(h264mi_decoder *dec);
/* Wait for the stream and collect per-slice status written by the entropy kernels. */
int32_t h264mi_batch_sync(h264mi_decoder *dec);
/* prepare + execute + sync */
int32_t h264mi_decode_batch(h264mi_decoder *dec, int32_t n_streams, const uint8_t *const *bufs, const size_t *lens, h264mi_batch_info *info);

/* Frames of the last batch, in decoding order (== output order unless the stream has B pictures: h264mi_stream_output_order).  A frame coded as two
 * field pictures (h264/slice.go:867-872) is ONE frame here: it is reported with the batch that holds its second field -- the first field's batch
 * reports nothing for it --, or, if the second field never comes, with the batch in which something else follows it (another picture, an
 * end-of-sequence / end-of-stream NAL unit), the rows of the missing field mid-grey. */
int32_t h264mi_stream_frame_count(h264mi_decoder *dec, int32_t stream, int32_t *n);
/* Device pointers + pitches of a decoded frame (coded size; planes are resident in HBM until the
 * next h264mi_batch_prepare). */
int32_t h264mi_frame_device_planes(h264mi_decoder *dec, int32_t stream, int32_t frame, void **y, void **cb, void **cr, int32_t *pitch_y,
                                   int32_t *pitch_c, int32_t *coded_width, int32_t *coded_height);
/* Geometry and picture-order data of a decoded frame (the picture's own SPS: a batch may span a resolution change). */
typedef struct {
    int32_t width, height, coded_width, coded_height, crop_x, crop_y; /* display size, coded size, crop origin (luma samples) */
    int32_t pic_order_cnt;                                            /* PicOrderCnt(CurrPic) 8.2.1, after a possible MMCO 5 */
    int32_t frame_num, nal_ref_idc, idr;
    int32_t new_sequence; /* picture order counts start over here: IDR picture, or memory_management_control_operation 5 */
} h264mi_frame_info;
int32_t h264mi_frame_get_info(h264mi_decoder *dec, int32_t stream, int32_t frame, h264mi_frame_info *info);
/* Output (display) order of the frames of the last batch of one stream: order[k] = index (decoding order) of the k-th frame
 * to show -- ascending PicOrderCnt inside each coded video sequence (a new one starts at an IDR picture or at a picture
 * with memory_management_control_operation 5).  The batch is ordered on its own: a caller that cuts batches in the middle
 * of a group of B pictures merges the tail of one batch with the head of the next by pic_order_cnt.  (The reference has no
 * output process at all: h264/server.go:113-166 stops at the parsed slice.) */
int32_t h264mi_stream_output_order(h264mi_decoder *dec, int32_t stream, int32_t *order, int32_t cap, int32_t *n);
/* Copy a frame to host memory as tight I420 (crop != 0: display size, else coded size). */
int32_t h264mi_frame_read(h264mi_decoder *dec, int32_t stream, int32_t frame, int32_t crop, uint8_t *dst, size_t cap);
/* Cropped, tightly packed I420 copy on the device (K6): dst is a DEVICE pointer. */
int32_t h264mi_frame_pack_device(h264mi_decoder *dec, int32_t stream, int32_t frame, void *dst_device, size_t cap);
/* The same for every frame of the last executed batch in ONE launch: frames of stream `stream` (or of all streams when
 * stream = -1, stream-major) in decoding order, back to back.  *bytes receives the total size (also on H264MI_ECAPACITY). */
int32_t h264mi_batch_pack_device(h264mi_decoder *dec, int32_t stream, void *dst_device, size_t cap, size_t *bytes);

/* Debug / test access to the intermediate macroblock records of a frame (host copy).
 * rec: 128 bytes per MB (layout: h264decode_amd/csrc/mi_types.h struct MbRec). */
int32_t h264mi_frame_read_mbrecs(h264mi_decoder *dec, int32_t stream, int32_t frame, uint8_t *rec, size_t cap);
/* ... and to the list-1 motion vectors of a picture with B slices: 64 bytes per MB, int16 (x, y) per 4x4 block in raster order
 * (zeros for pictures without B slices).  Together they are what h264/slice.go:77-102 SliceData holds per macroblock. */
int32_t h264mi_frame_read_mbmv1(h264mi_decoder *dec, int32_t stream, int32_t frame, uint8_t *mv1, size_t cap);

/* Time (ms) spent by the kernels of the last execute, measured with HIP events on the decoder's
 * stream: [0] entropy, [1] inter recon, [2] intra recon, [3] deblock, [4] total.  Valid after sync
 * when profiling was enabled with h264mi_decoder_set_profiling(dec, 1). */
int32_t h264mi_decoder_set_profiling(h264mi_decoder *dec, int32_t on);
int32_t h264mi_last_kernel_times(h264mi_decoder *dec, double ms[5]);
/* Duration (ms) of every single launch of one kernel in that pass, in launch order (kernel: 0 entropy, 1 inter, 2 intra,
 * 3 deblock; launch k of the pixel kernels handles picture k of every stream, so launch 0 of a GOP is the IDR picture).
 * *n receives the number of launches; at most cap values are written. */
int32_t h264mi_last_launch_times(h264mi_decoder *dec, int32_t kernel, float *ms, int32_t cap, int32_t *n);

/* Device memory (bytes) the decoder holds right now: everything is sized at create time, except what only B pictures need
 * (list-1 vectors, co-located motion arrays), which is allocated when a stream's first B slice arrives. */
int32_t h264mi_decoder_memory(h264mi_decoder *dec, int64_t *device_bytes);

/* Residual-coefficient pool: blocks (32 bytes each) the fullest of the pipelined batches still on the device took, and the
 * capacity.  Large decoders reserve 8 blocks per macroblock, not the worst case of 26 (a batch that needs more fails with
 * H264MI_EDECODE, "code 40"); the environment variable H264MI_COEF_BLOCKS_PER_MB, read at create time, sizes it.  Call after
 * h264mi_batch_sync. */
int32_t h264mi_decoder_coef_pool(h264mi_decoder *dec, int64_t *used_blocks, int64_t *capacity_blocks);
/* Slices of CABAC field pictures (h264mi_config.allow_unpinned_field_cabac) that failed in the entropy kernels since the decoder was created: what a wrong
 * value in the unpinned context tables of field-coded blocks looks like (a damaged stream looks the same). */
int32_t h264mi_decoder_unpinned_failures(h264mi_decoder *dec, int64_t *n);

const char *h264mi_last_error_string(void);
const char *h264mi_version(void);

/* Exported but NOT part of the ABI (test hooks of this repository's own suite, may change or vanish): h264mi_internal_poison,
 * h264mi_internal_deblock_plan, h264mi_internal_band_plan, h264mi_internal_deblock_phase_clocks, h264mi_internal_vlc_selftest. */

#ifdef __cplusplus
}
#endif
#endif

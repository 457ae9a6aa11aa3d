// h264decode_amd/csrc/mi_api.cpp -- the C ABI of libh264mi.so (include/h264mi.h): host-side
// front end (NAL dispatch h264/server.go:113-166, picture management 8.2) and GPU launch sequence.
//
// There is deliberately NO CPU pixel path in this library: every sample is produced by the HIP
// kernels in k_entropy.hip / k_recon.hip / k_deblock.hip.  If no device is usable the create call
// fails (H264MI_ENODEVICE).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <thread>
#include <functional>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/h264mi.h"
#include "mi_kernels.h"
#include "mi_parse.hpp"
#include "mi_tables.h"

using namespace mi;

#ifndef MI_SETS
#define MI_SETS 3 /* record / coefficient sets: a pass may run two passes ahead of the reconstruction that frees its set (a fourth set bought 1 % for 28 GB) */
#endif

#define HIP_TRY(x)                                                                          \
    do {                                                                                    \
        hipError_t _e = (x);                                                                \
        if (_e != hipSuccess) {                                                             \
            set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
            return H264MI_EDEVICE;                                                          \
        }                                                                                   \
    } while (0)

// ---------------------------------------------------------------- device tables
static void put_vlc(uint16_t *lut, int bits, int len, uint32_t code, uint16_t value) {
    if (!len || len > bits) return;
    uint32_t base = code << (bits - len), n = 1u << (bits - len);
    for (uint32_t i = 0; i < n; i++) lut[base + i] = static_cast<uint16_t>((len << 8) | value);
}
// The CAVLC tables: direct-indexed first (every window of L bits -> len << 8 | value, from the code lists of mi_tables.h), then folded into the compact
// form the kernels keep in LDS (mi_types.h: MI_VLC_*).  `mismatches` (h264mi_internal_vlc_selftest): every window of every direct table is looked
// up in the compact one as the kernels do it, and run_before's closed form for zerosLeft > 6 is compared with its table.
static void build_vlc(uint16_t *c, int *mismatches) {
    std::vector<uint16_t> ct[3] = {std::vector<uint16_t>(1u << MI_VLC_CT0_L), std::vector<uint16_t>(1u << MI_VLC_CT1_L), std::vector<uint16_t>(1u << MI_VLC_CT2_L)};
    std::vector<uint16_t> ct3(64), cdc(1u << MI_VLC_CDC_L), tz(15u << MI_VLC_TZ_L), cdctz(24), run(7u << 11);
    const int ctl[3] = {MI_VLC_CT0_L, MI_VLC_CT1_L, MI_VLC_CT2_L}, ctbase[3] = {MI_VLC_CT0, MI_VLC_CT1, MI_VLC_CT2};
    for (int tc = 0; tc <= 16; tc++)
        for (int t1 = 0; t1 <= std::min(tc, 3); t1++) {
            const uint16_t v = static_cast<uint16_t>((tc << 2) | t1);
            for (int k = 0; k < 3; k++) put_vlc(ct[k].data(), ctl[k], mi_coeff_token_len[k][4 * tc + t1], mi_coeff_token_bits[k][4 * tc + t1], v);
            put_vlc(ct3.data(), 6, mi_coeff_token_len[3][4 * tc + t1], mi_coeff_token_bits[3][4 * tc + t1], v);
            if (tc <= 4) put_vlc(cdc.data(), MI_VLC_CDC_L, mi_chroma_dc_token_len[4 * tc + t1], mi_chroma_dc_token_bits[4 * tc + t1], v);
        }
    for (int tc = 1; tc <= 15; tc++)
        for (int z = 0; z <= 16 - tc && z < 16; z++)
            put_vlc(tz.data() + ((tc - 1) << MI_VLC_TZ_L), MI_VLC_TZ_L, mi_total_zeros_len[tc - 1][z], mi_total_zeros_bits[tc - 1][z], static_cast<uint16_t>(z));
    for (int tc = 1; tc <= 3; tc++)
        for (int z = 0; z <= 4 - tc; z++) put_vlc(cdctz.data() + 8 * (tc - 1), 3, mi_chroma_dc_total_zeros_len[tc - 1][z], mi_chroma_dc_total_zeros_bits[tc - 1][z], static_cast<uint16_t>(z));
    for (int zl = 1; zl <= 7; zl++)
        for (int r = 0; r < 15; r++) put_vlc(run.data() + ((zl - 1) << 11), 11, mi_run_len[zl - 1][r], mi_run_bits[zl - 1][r], static_cast<uint16_t>(r));
    memset(c, 0, sizeof(uint16_t) * MI_VLC_N);
    int bad = 0;
    // fold: the entry of group lz, suffix s is what the direct table says for the window 0^lz 1 s 0...; the code must end inside those bits
    auto fold = [&](const uint16_t *direct, int L, int S, uint16_t *out) {
        for (int lz = 0; lz <= L; lz++)
            for (uint32_t sfx = 0; sfx < (1u << S); sfx++) {
                uint32_t win = lz < L ? (1u << (31 - lz)) : 0u; // as a 32-bit window, MSB first
                if (lz + 1 < 32) win |= lz < L ? (sfx << (32 - S)) >> (lz + 1) : 0u;
                const uint16_t e = direct[win >> (32 - L)];
                if ((e >> 8) > lz + 1 + S && lz < L) bad++; // a code longer than the bits that select its entry
                out[(lz << S) | sfx] = e;
            }
    };
    for (int k = 0; k < 3; k++) fold(ct[k].data(), ctl[k], MI_VLC_CT_S, c + ctbase[k]);
    memcpy(c + MI_VLC_CT3, ct3.data(), 64 * sizeof(uint16_t));
    fold(cdc.data(), MI_VLC_CDC_L, MI_VLC_CDC_S, c + MI_VLC_CDC);
    for (int k = 0; k < 15; k++) fold(tz.data() + (k << MI_VLC_TZ_L), MI_VLC_TZ_L, MI_VLC_TZ_S, c + MI_VLC_TZ + k * MI_VLC_TZ_STRIDE);
    memcpy(c + MI_VLC_CDCTZ, cdctz.data(), 24 * sizeof(uint16_t));
    for (int zl = 1; zl <= 6; zl++)
        for (int i = 0; i < 8; i++) c[MI_VLC_RUN + 8 * (zl - 1) + i] = run[((zl - 1) << 11) + (i << 8)];
    if (!mismatches) return;
    // every window of every direct table, looked up the way the kernels do it
    auto check = [&](const uint16_t *direct, int L, int S, const uint16_t *folded) {
        for (uint32_t i = 0; i < (1u << L); i++) {
            const uint32_t w = i << (32 - L);
            const int lz = w ? __builtin_clz(w) : 32;
            if (folded[MI_VLC_INDEX(w, lz, L, S)] != direct[i]) bad++;
        }
    };
    for (int k = 0; k < 3; k++) check(ct[k].data(), ctl[k], MI_VLC_CT_S, c + ctbase[k]);
    check(cdc.data(), MI_VLC_CDC_L, MI_VLC_CDC_S, c + MI_VLC_CDC);
    for (int k = 0; k < 15; k++) check(tz.data() + (k << MI_VLC_TZ_L), MI_VLC_TZ_L, MI_VLC_TZ_S, c + MI_VLC_TZ + k * MI_VLC_TZ_STRIDE);
    for (int zl = 1; zl <= 6; zl++)
        for (uint32_t i = 0; i < 2048; i++) {
            const uint16_t e = run[((zl - 1) << 11) + i];
            if (e && c[MI_VLC_RUN + 8 * (zl - 1) + (i >> 8)] != e) bad++; // (codes of at most 3 bits; e == 0: a window no code matches)
            if (!e && c[MI_VLC_RUN + 8 * (zl - 1) + (i >> 8)] != 0) bad++;
        }
    for (uint32_t i = 0; i < 2048; i++) {
        const uint32_t w = i << 21;
        const int lz = w ? __builtin_clz(w) : 32;
        if (MI_RUN_BEFORE_LONG(w, lz) != run[(6u << 11) + i]) bad++;
    }
    *mismatches = bad;
}
#if defined(H264MI_TEST_HOOKS)
extern "C" int32_t h264mi_internal_vlc_selftest(void) {
    std::vector<uint16_t> c(MI_VLC_N);
    int bad = -1;
    build_vlc(c.data(), &bad);
    return bad;
}
#endif
static void build_tables(DevTables *t) {
    memset(t, 0, sizeof(*t));
    memcpy(t->range_lps, mi_range_lps, sizeof(t->range_lps));
    memcpy(t->trans_lps, mi_trans_lps, sizeof(t->trans_lps));
    // 9.3.1.1 (9-5): h264/cabac.go:118-121 PreCtxState, :158-164 split into pStateIdx / valMPS
    for (int set = 0; set < 4; set++)
        for (int qp = 0; qp < 52; qp++)
            for (int i = 0; i < MI_NCTX; i++) {
                int pre = ((mi_cabac_mn[set][i][0] * qp) >> 4) + mi_cabac_mn[set][i][1];
                pre = std::min(std::max(pre, 1), 126);
                // pStateIdx in bits 0..5, valMPS in bit 6: the state is a lane number as it stands (v_readlane takes its select modulo 64)
                t->ctx_init[set][qp][i] = pre <= 63 ? static_cast<uint8_t>(63 - pre) : static_cast<uint8_t>((pre - 64) | 64);
            }
    memcpy(t->sig8x8, mi_sig8x8_ctx, 63);
    memcpy(t->sig8x8_field, mi_sig8x8_field_ctx, 63);
    memcpy(t->last8x8, mi_last8x8_ctx, 63);
    memcpy(t->zigzag4, mi_zigzag4x4, 16);
    memcpy(t->zigzag8, mi_zigzag8x8, 64);
    memcpy(t->fieldscan4, mi_fieldscan4x4, 16);
    memcpy(t->fieldscan8, mi_fieldscan8x8, 64);
    memcpy(t->me_intra, mi_me_intra, 48), memcpy(t->me_intra + 48, mi_me_intra0, 16);
    memcpy(t->me_inter, mi_me_inter, 48), memcpy(t->me_inter + 48, mi_me_inter0, 16);
    memcpy(t->alpha, mi_alpha, 52);
    memcpy(t->beta, mi_beta, 52);
    for (int i = 0; i < 52; i++) {
        for (int b = 0; b < 3; b++) t->tc0[i][b + 1] = mi_tc0[i][b];
        t->qpc[i] = i < 30 ? static_cast<uint8_t>(i) : mi_qpc_tab[i - 30];
    }
    build_vlc(t->vlc_c, nullptr);
}
static void build_scaling(const uint8_t s4[6][16], const uint8_t s8[2][64], ScalingSet *o) { // 8.5.9
    for (int l = 0; l < 6; l++)
        for (int q = 0; q < 6; q++)
            for (int k = 0; k < 16; k++) {
                int r = mi_zigzag4x4[k], x = r & 3, y = r >> 2;
                int v = (!(x & 1) && !(y & 1)) ? mi_norm4x4[q][0] : (((x & 1) && (y & 1)) ? mi_norm4x4[q][1] : mi_norm4x4[q][2]);
                o->ls4[l][q][r] = static_cast<uint16_t>(s4[l][k] * v);
            }
    for (int l = 0; l < 2; l++)
        for (int q = 0; q < 6; q++)
            for (int k = 0; k < 64; k++) {
                int r = mi_zigzag8x8[k], x = r & 7, y = r >> 3, c;
                if (!(x & 3) && !(y & 3))
                    c = 0;
                else if ((x & 1) && (y & 1))
                    c = 1;
                else if ((x & 3) == 2 && (y & 3) == 2)
                    c = 2;
                else if ((!(y & 3) && (x & 1)) || ((y & 1) && !(x & 3)))
                    c = 3;
                else if ((!(y & 3) && (x & 3) == 2) || ((y & 3) == 2 && !(x & 3)))
                    c = 4;
                else
                    c = 5;
                o->ls8[l][q][r] = static_cast<uint16_t>(s8[l][k] * mi_norm8x8[q][c]);
            }
}

// ---------------------------------------------------------------- per-stream host state
struct Slot {
    int ref = 0; // 0 unused, 1 short-term, 2 long-term
    int frame_num = 0, frame_num_wrap = 0, pic_num = 0, long_idx = 0, poc = 0;
    // Output of the current batch, or a reference picture at the start of the batch: not reused before the next prepare.
    // (The second half keeps h264mi_batch_execute repeatable: a slot freed by a marking operation in the middle of the
    // batch still holds the samples earlier pictures of the batch predict from.)
    bool held = false;
    bool nonexisting = false; // a frame inferred by the gaps-in-frame_num process (8.2.5.2): a place in the window, no picture
    int pic = -1; // index into the PicDesc table of the batch being prepared, -1: decoded by an earlier batch (field-coded frames: fpic[])
    // Fields (h264/slice.go:867-872 field_pic_flag / bottom_field_flag; h264/sps.go:316-322).  A frame slot holds both fields of a frame, however
    // they were coded: a frame picture fills both at once (fields = 3), a field picture the rows of its parity.
    int fields = 0;          // decoded fields: bit 0 top, bit 1 bottom
    int funref = 0;          // fields taken out of the reference set one by one (memory_management_control_operation 1 in a field picture, 8.2.5.4.1)
    int fpoc[2] = {0, 0};    // PicOrderCnt of the top / bottom field (8.2.1); `poc` is the frame's: the smaller one, or that of the only field there is
    int fpic[2] = {-1, -1};  // PicDesc of the field pictures decoded by the batch being prepared
    bool field_coded = false;          // coded as field pictures (direct prediction needs a co-located picture of the same structure as the current one)
    bool col_valid[2] = {false, false}; // the ColRec array of the frame / top field [0], the bottom field [1] holds this picture's motion
};
struct OutFrame { // a decoded picture of the current batch, with the geometry it was coded with
    int slot, wmb, hmb, crop_x, crop_y, width, height;
    int poc, frame_num, nal_ref_idc, idr, pic;
    int new_sequence; // IDR picture or memory_management_control_operation 5: picture order counts start over
    int pic2 = -1;    // a frame coded as two field pictures: `pic` / `pic2` are the first / second field's PicDesc (-1: decoded by an earlier batch, or never)
};
struct StreamState {
    h264mi_sps sps[32];
    h264mi_pps pps[256];
    std::vector<uint8_t> sg_ids[256]; // slice_group_id[] of the PPSs with slice_group_map_type 6 (h264/pps.go:23)
    std::vector<int32_t> cur_first_mbs; // first_mb_in_slice of the slices of the current picture
    uint32_t epoch = 0;                 // bumped by every reset of the stream: failures of batches prepared before it are nobody's business any more
    bool sps_ok[32] = {}, pps_ok[256] = {};
    int active_sps = -1;
    int wmb = 0, hmb = 0;
    std::vector<Slot> slots;
    int prev_poc_msb = 0, prev_poc_lsb = 0, prev_frame_num = 0, prev_frame_num_offset = 0;
    int top_above_poc = 0; // TopFieldOrderCnt - PicOrderCnt of the picture compute_poc() was last asked about (> 0: its bottom field comes first)
    int poc_top = 0, poc_bot = 0; // TopFieldOrderCnt / BottomFieldOrderCnt of that picture (a field picture: both its one count)
    int prev_ref_frame_num = 0; // PrevRefFrameNum (7.4.3): frame_num of the previous reference picture; 0 after an IDR picture or operation 5
    // picture under construction
    int cur_slot = -1, cur_pic = -1, cur_slices = 0;
    int cur_field = 0;        // the picture under construction is 0 a frame, 1 a top field, 2 a bottom field
    bool cur_second = false;  // ... and the second field of its frame (it may predict from the first one)
    // A frame whose first field has been decoded waits here for its second field (the next picture, if it is a field of the other parity
    // with the same frame_num, 7.4.1.2.4 / 3.30); it goes out -- once -- when that field is complete, or with one field decoded (the rows of
    // the other one mid-grey) when something else follows: another picture, an end-of-sequence / end-of-stream NAL unit, a reset.  The wait
    // may span a batch boundary.
    int pend_slot = -1;
    OutFrame pend_out;
    h264mi_slice_header first_sh;
    int n_pics_in_batch = 0;
    int status = H264MI_OK; // of this stream in the current batch (h264mi_stream_status)
    const void *status_batch = nullptr; // the batch whose entropy kernels set `status` (harvest_status)
    bool need_idr = false;  // after an error: nothing is decodable before the next IDR picture
};

// Everything one prepared batch owns: the pinned staging buffers and their device mirrors, the descriptors, the launch
// lists and the output frame lists.  There are MI_STAGES of them, so that h264mi_batch_prepare(n + 1) -- host parsing and
// the H2D copies -- runs while batch n is still executing (h264/server.go:144-145 reads its connection in an endless loop).
#ifndef MI_STAGES
#define MI_STAGES 2
#endif
struct Stage {
    uint8_t *d_bits = nullptr, *h_bits = nullptr;
    size_t bits_used = 0;
    size_t map_cursor = 0, bits_end = 0; // slice group maps of FMO pictures follow the slices in the staging buffer; bits_end: what must be uploaded
    std::vector<uint32_t> fmo_pics;      // pictures whose records are zeroed before the entropy kernels run
    struct GreyFill { uint32_t stream, slot, parity, w, h; }; // a frame that went out with one field decoded: the rows of the other parity are painted mid-grey
    std::vector<GreyFill> grey;
    std::vector<uint32_t> epochs;        // StreamState::epoch of every stream when this batch was prepared
    SliceDesc *d_slices = nullptr, *h_slices = nullptr;
    PicDesc *d_pics = nullptr, *h_pics = nullptr;
    uint32_t *d_status = nullptr, *h_status = nullptr, *d_lists = nullptr, *h_lists = nullptr;
    BSliceExt *d_bext = nullptr, *h_bext = nullptr; // one per B slice
    int n_bext = 0;
    // B pictures take their direct-mode motion from RefPicList1[0], so that picture's slices must have been entropy-decoded
    // first: slices are launched by level -- 0: I and P slices (k_entropy), n: B slices whose co-located picture is of level
    // n - 1 or older than the batch (k_entropy_b) -- and the slice table is ordered by level.
    std::vector<int> pic_level, slice_level; // per picture / per slice of the batch (slice_level in parse order, until the table is sorted)
    std::vector<uint8_t> pic_save_col;       // the picture's motion is kept for later direct prediction (k_dbprep writes its ColRec array)
    std::vector<int> pic_wave;               // reconstruction wave of the picture: 0 for a picture that reads no picture of this batch, else 1 + the latest wave among its references
    std::vector<int> level_first;            // first slice of each level in the sorted table (+ end marker)
    std::vector<uint32_t> colsave_n;            // per level: pictures whose ColRec array k_dbprep writes (what the cross-pass fence looks at)
    std::vector<uint32_t> prep_off, prep_n;     // per level: the pictures whose last slice is of that level (k_dbprep runs on them after the level), as a range of d_lists
    std::vector<uint32_t> wave_b_off, wave_b_n, wave_p_off, wave_p_n, wave_nb_off, wave_nb_n; // per wave: B pictures / inter non-B / non-B
    int n_slices = 0, n_pics = 0, wmb_max = 0, hmb_max = 0, mbs_max = 0;
    uint64_t mb_used = 0;
    std::vector<std::vector<uint32_t>> waves, waves_inter;
    std::vector<uint32_t> wave_off, wave_inter_off;
    std::vector<std::vector<OutFrame>> out; // per stream: frames of this batch in decoding order
    bool prepared = false, executed = false;
    bool harvested = true; // the entropy kernels' per-slice status words of the last execute have been looked at (harvest_status)
    bool retried = false;  // the batch has been repeated with the whole residual pool (retry_exhausted)
    h264mi_batch_info info;
    hipEvent_t ev_upload = nullptr; // H2D copies of this batch are complete
    hipEvent_t ev_done = nullptr;   // the last pass over this batch has finished (nothing reads or writes its buffers any more)
};

struct h264mi_decoder {
    h264mi_config cfg;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::vector<StreamState> st;
    int Wmax = 0, Hmax = 0, n_slots = 0;
    size_t slot_bytes = 0;
    Stage stage[MI_STAGES];
    int prep = 0;  // stage of the most recent h264mi_batch_prepare
    int exec = 0;  // stage of the most recent h264mi_batch_execute: what sync and the frame accessors refer to
    size_t bits_cap = 0;
    uint32_t *d_toprows[MI_SETS] = {}; // entropy kernels' row-above neighbour state: 48 B per MB column per slice
    int slices_cap = 0, pics_cap = 0;
    // Passes are pipelined: the entropy kernels of passes n+1 / n+2 (two private streams, alternating)
    // overlap the reconstruction kernels of pass n (on `stream`); each pass owns one of MI_SETS
    // MbRec / coefficient buffer sets, fenced by events.
    MbRec *d_mbrec[MI_SETS] = {};
    DbPrm *d_dbprm[MI_SETS] = {};        // k_dbprep -> K5: boundary strengths and filter parameters, same indexing as d_mbrec
    unsigned long long *d_imask[MI_SETS] = {}; // k_dbprep -> K3: one bit per macroblock of the batch (same indexing), set for what K3 reconstructs
    MbMv1 *d_mv1[MI_SETS] = {};          // list-1 vectors, same indexing as d_mbrec; allocated when the first B slice arrives
    ColRec *d_colrec = nullptr;          // per stream and frame slot: the motion a picture leaves for later direct prediction
    size_t colrec_per_slot = 0;          // ColRecs per frame slot
    int16_t *d_coef[MI_SETS] = {};       // coefficient pools: 32-byte blocks, only the blocks that carry anything (MbRec::coef_off / coef_mask)
    uint32_t *d_pool_head = nullptr;     // MI_SETS counters: next free block of each pool, reset before every entropy launch
    uint64_t pool_blocks = 0;            // blocks per pool
    hipStream_t ent_stream[2] = {nullptr, nullptr};
    hipStream_t rec_stream = nullptr; // K3-K5; the caller's stream only brackets a pass with events
    hipEvent_t ev_user = nullptr;
    hipEvent_t ev_ent[MI_SETS] = {}, ev_rec[MI_SETS] = {}, ev_col[MI_SETS] = {};
    uint64_t pass = 0; // execute() counter
    int last_exec_stage = -1, last_exec_set = 0; // staging set and record set of the most recent execute (ensure_b_buffers)
    bool last_pass_had_b = false;
    uint64_t mb_cap = 0;
    uint64_t dev_bytes = 0;              // device memory this decoder holds (h264mi_decoder_memory)
    uint32_t *d_backfill = nullptr;      // picture list of the one-off ColRec back-fill (first B slice of a decoder)
    FramePool *d_pools = nullptr;
    std::vector<FramePool> h_pools;
    uint8_t *d_frames = nullptr;
    DevTables *d_tables = nullptr, *h_tables = nullptr;
    int n_scaling = 0;
    bool tables_dirty = true;
    size_t ent_lds_pad = 0; // dynamic LDS requested (and not used) by k_entropy: caps its wavefronts per CU, see h264mi_decoder_create
    bool isolate = false; // h264mi_decoder_set_isolation: a broken stream does not fail the batch
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> ev_kind;
    size_t ev_used = 0;
    double k_ms[5] = {0, 0, 0, 0, 0};
    // Launches with fewer pictures than the chip has CUs spread a picture over several workgroups (k_intra_x, k_deblock_x,
    // k_deblock_x): hand-off rings / flags in global memory, a ticket counter per kernel family, a give-up word
    unsigned long long *d_xring = nullptr; // K5: x_cap * (Wmax / 16) * 24 granules
    uint32_t *d_xdone = nullptr;           // K3: x_cap * (Wmax / 16) flag words
    int k5_max_waves = MI_DEBLOCK8_MAX_WAVES;
    int64_t unpinned_failures = 0; // slices of CABAC field pictures that failed in the entropy kernel (h264mi_decoder_unpinned_failures)
    uint32_t *d_xctl = nullptr, *h_xstatus = nullptr; // [0] K5 tickets, [32] K3 tickets, [64] give-up code (128-byte lines of their own)
    uint32_t x_epoch = 0, x_tk5 = 0, x_tk3 = 0;
    int x_max_wgs = 256, x_cap = 512, x_cap3 = 512; // workgroups per launch: default; capacity of the K5 ring; of the K3 flag array
    // K6 descriptor tables: two of them, used alternately, each fenced by an event -- a pack call does not wait for the stream (the batch it packs
    // may still be executing, and the next execute may be enqueued right behind it)
    PackDesc *h_pack[2] = {nullptr, nullptr}, *d_pack[2] = {nullptr, nullptr};
    size_t pack_cap[2] = {0, 0};
    hipEvent_t ev_pack[2] = {nullptr, nullptr};
    int pack_slot = 0;
    hipEvent_t last_pack = nullptr; // the most recent pack launch: the reconstruction of a later pass may rewrite the frames it reads
    std::vector<float> launch_ms[4]; // duration of every launch of the last profiled pass, per kernel
};

static int g_device = -1;

// hipSetDevice is per-thread state: every entry point that takes a decoder selects the decoder's device for the
// duration of the call and restores the caller's afterwards (callers may be pool threads or migrating goroutines).
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(const h264mi_decoder *d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != d->cfg.device) ok = hipSetDevice(d->cfg.device) == hipSuccess;
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define GUARD(d)                                                        \
    DeviceGuard guard_(d);                                              \
    if (!guard_.ok) {                                                   \
        set_error("hipSetDevice(%d) failed", (d)->cfg.device);          \
        return H264MI_EDEVICE;                                          \
    }

extern "C" const char *h264mi_last_error_string(void) { return last_error(); }
extern "C" const char *h264mi_version(void) { return "h264mi 0.1 (gfx950)"; }

extern "C" int32_t h264mi_annexb_scan(const uint8_t *buf, size_t len, h264mi_nal *out, int32_t cap, int32_t *n) {
    if (!buf || !out || !n || cap < 0) return H264MI_EINVAL;
    int cnt = 0;
    int r = annexb_scan(buf, len, out, cap, &cnt);
    *n = cnt;
    return r;
}
extern "C" int32_t h264mi_nal_parse(const uint8_t *nal, size_t len, h264mi_nal *hdr, uint8_t *rbsp, size_t *rbsp_len) { return nal_parse(nal, len, hdr, rbsp, rbsp_len); }
extern "C" int32_t h264mi_sps_parse(const uint8_t *rbsp, size_t len, h264mi_sps *sps) { return parse_sps(rbsp, len, sps); }
extern "C" int32_t h264mi_pps_parse(const h264mi_sps *sps, const uint8_t *rbsp, size_t len, h264mi_pps *pps) { return parse_pps(sps, rbsp, len, pps); }
extern "C" int32_t h264mi_pps_slice_group_ids(const h264mi_sps *sps, const uint8_t *rbsp, size_t len, uint8_t *ids, size_t cap, size_t *n) {
    h264mi_pps tmp;
    return parse_pps_ids(sps, rbsp, len, &tmp, ids, cap, n);
}
extern "C" int32_t h264mi_map_unit_to_slice_group_map(const h264mi_sps *sps, const h264mi_pps *pps, const uint8_t *ids, size_t n_ids, int32_t cycle, uint8_t *map, size_t cap,
                                                      size_t *n) {
    return map_unit_to_slice_group_map(sps, pps, ids, n_ids, cycle, map, cap, n);
}
extern "C" int32_t h264mi_mb_to_slice_group_map(const h264mi_sps *sps, const h264mi_pps *pps, const uint8_t *ids, size_t n_ids, int32_t cycle, int32_t field_pic, uint8_t *map,
                                                size_t cap, size_t *n) {
    return mb_to_slice_group_map(sps, pps, ids, n_ids, cycle, field_pic, map, cap, n);
}
extern "C" int32_t h264mi_next_mb_address(const uint8_t *map, size_t n_mbs, size_t n) { return next_mb_address(map, n_mbs, n); }
extern "C" int32_t h264mi_slice_header_parse(const h264mi_sps *sps, const h264mi_pps *pps, int32_t nal_ref_idc, int32_t nal_unit_type, const uint8_t *rbsp,
                                             size_t len, h264mi_slice_header *sh) {
    return parse_slice_header(sps, pps, nal_ref_idc, nal_unit_type, rbsp, len, sh);
}

extern "C" int32_t h264mi_init(int32_t device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device available: the decode path needs a gfx950 GPU (there is no CPU fallback)");
        return H264MI_ENODEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (0..%d)", device, n - 1);
        return H264MI_EINVAL;
    }
    HIP_TRY(hipSetDevice(device));
    g_device = device;
    return H264MI_OK;
}

static void free_all(h264mi_decoder *d) {
    for (Stage &g : d->stage) {
        if (g.d_bits) hipFree(g.d_bits);
        if (g.h_bits) hipHostFree(g.h_bits);
        if (g.d_slices) hipFree(g.d_slices);
        if (g.h_slices) hipHostFree(g.h_slices);
        if (g.d_pics) hipFree(g.d_pics);
        if (g.h_pics) hipHostFree(g.h_pics);
        if (g.d_status) hipFree(g.d_status);
        if (g.h_status) hipHostFree(g.h_status);
        if (g.d_lists) hipFree(g.d_lists);
        if (g.h_lists) hipHostFree(g.h_lists);
        if (g.d_bext) hipFree(g.d_bext);
        if (g.h_bext) hipHostFree(g.h_bext);
        if (g.ev_upload) hipEventDestroy(g.ev_upload);
        if (g.ev_done) hipEventDestroy(g.ev_done);
    }
    for (int i = 0; i < MI_SETS; i++) {
        if (d->d_mbrec[i]) hipFree(d->d_mbrec[i]);
        if (d->d_mv1[i]) hipFree(d->d_mv1[i]);
        if (d->d_dbprm[i]) hipFree(d->d_dbprm[i]);
        if (d->d_imask[i]) hipFree(d->d_imask[i]);
        if (i == 0 && d->d_coef[0]) hipFree(d->d_coef[0]);
        if (i == 0 && d->d_pool_head) hipFree(d->d_pool_head);
        if (d->d_toprows[i]) hipFree(d->d_toprows[i]);
        if (d->ev_ent[i]) hipEventDestroy(d->ev_ent[i]);
        if (d->ev_col[i]) hipEventDestroy(d->ev_col[i]);
        if (d->ev_rec[i]) hipEventDestroy(d->ev_rec[i]);
    }
    for (int i = 0; i < 2; i++)
        if (d->ent_stream[i]) hipStreamDestroy(d->ent_stream[i]);
    if (d->rec_stream) hipStreamDestroy(d->rec_stream);
    if (d->ev_user) hipEventDestroy(d->ev_user);
    if (d->d_colrec) hipFree(d->d_colrec);
    if (d->d_backfill) hipFree(d->d_backfill);
    if (d->d_xring) hipFree(d->d_xring);
    if (d->d_xdone) hipFree(d->d_xdone);
    if (d->d_xctl) hipFree(d->d_xctl);
    if (d->h_xstatus) hipHostFree(d->h_xstatus);
    if (d->d_pools) hipFree(d->d_pools);
    for (int i = 0; i < 2; i++) {
        if (d->h_pack[i]) hipHostFree(d->h_pack[i]);
        if (d->d_pack[i]) hipFree(d->d_pack[i]);
        if (d->ev_pack[i]) hipEventDestroy(d->ev_pack[i]);
    }
    if (d->d_frames) hipFree(d->d_frames);
    if (d->d_tables) hipFree(d->d_tables);
    if (d->h_tables) hipHostFree(d->h_tables);
    for (auto e : d->ev) hipEventDestroy(e);
    if (d->own_stream && d->stream) hipStreamDestroy(d->stream);
}

static int ensure_b_buffers(h264mi_decoder *d);

extern "C" int32_t h264mi_decoder_create(const h264mi_config *cfg_, h264mi_decoder **out) {
    if (!cfg_ || !out) return H264MI_EINVAL;
    // the caller's struct may be shorter (an older header) or longer (a newer one) than this build's: fields beyond either end are 0
    if (cfg_->struct_size < offsetof(h264mi_config, max_ref_frames) || cfg_->struct_size > 4096) {
        set_error("h264mi_decoder_create: h264mi_config.struct_size = %u (set it to sizeof(h264mi_config); the struct must be zero-initialised)", cfg_->struct_size);
        return H264MI_EINVAL;
    }
    h264mi_config cfg_copy;
    memset(&cfg_copy, 0, sizeof(cfg_copy));
    memcpy(&cfg_copy, cfg_, std::min<size_t>(cfg_->struct_size, sizeof(cfg_copy)));
    const h264mi_config *cfg = &cfg_copy;
    if (cfg->max_streams < 1 || cfg->max_width < 16 || cfg->max_height < 16 || cfg->max_frames_per_batch < 1) return H264MI_EINVAL;
    int r = h264mi_init(cfg->device);
    if (r != H264MI_OK) return r;
    h264mi_decoder *d = new h264mi_decoder();
    d->cfg = *cfg;
    if (d->cfg.max_slices_per_frame < 1) d->cfg.max_slices_per_frame = 1;
    d->Wmax = (cfg->max_width + 15) & ~15;
    d->Hmax = (cfg->max_height + 15) & ~15;
    if (d->Wmax > 8192 || d->Hmax > 5120) { // LDS row-state of K3 (320 rows x 512 columns of macroblocks) and K5 (96 row groups)
        set_error("h264mi_decoder_create: pictures larger than 8192x5120 are not supported");
        delete d;
        return H264MI_EUNSUPPORTED;
    }
    // frame slots per stream: the outputs of the batch being prepared and of the one before it (still being read or executed:
    // MI_STAGES batches are in flight), up to 16 reference pictures, and the picture under construction
    if (d->cfg.max_ref_frames < 1 || d->cfg.max_ref_frames > MI_MAX_REFS) d->cfg.max_ref_frames = MI_MAX_REFS;
    d->n_slots = MI_STAGES * cfg->max_frames_per_batch + d->cfg.max_ref_frames + 1;
    if (d->n_slots >= MI_REF_PARITY) { // (reference entries of field pictures keep the field's parity in bit 14 of the slot number)
        set_error("h264mi_decoder_create: max_frames_per_batch %d is too large (at most %d)", cfg->max_frames_per_batch, (MI_REF_PARITY - 18) / MI_STAGES);
        delete d;
        return H264MI_EINVAL;
    }
    d->slot_bytes = (static_cast<size_t>(d->Wmax) * d->Hmax * 3 / 2 + 255) & ~static_cast<size_t>(255);
    const int S = cfg->max_streams;
    d->st.resize(S);
    for (auto &s : d->st) s.slots.resize(d->n_slots);
    d->pics_cap = S * cfg->max_frames_per_batch;
    d->slices_cap = d->pics_cap * d->cfg.max_slices_per_frame;
    d->mb_cap = static_cast<uint64_t>(d->pics_cap) * (d->Wmax / 16) * (d->Hmax / 16);
    d->bits_cap = cfg->max_bitstream_bytes > 0 ? static_cast<size_t>(cfg->max_bitstream_bytes) : std::max<size_t>(static_cast<size_t>(d->pics_cap) * d->Wmax * d->Hmax / 2, 1 << 20);
    d->bits_cap = (d->bits_cap + 16 * static_cast<size_t>(d->slices_cap) + 8192 + 15) & ~static_cast<size_t>(15);
    auto fail = [&](int code) {
        free_all(d);
        delete d;
        return code;
    };
#define DEV_ALLOC(ptr, bytes)                      \
    do {                                           \
        const size_t _n = (bytes);                 \
        TRY_ALLOC(hipMalloc(&(ptr), _n));          \
        d->dev_bytes += _n;                        \
    } while (0)
#define TRY_ALLOC(x)                                                                 \
    do {                                                                             \
        hipError_t _e = (x);                                                         \
        if (_e != hipSuccess) {                                                      \
            set_error("%s failed: %s", #x, hipGetErrorString(_e));                   \
            return fail(_e == hipErrorOutOfMemory ? H264MI_ENOMEM : H264MI_EDEVICE); \
        }                                                                            \
    } while (0)
    if (cfg->hip_stream)
        d->stream = static_cast<hipStream_t>(cfg->hip_stream);
    else {
        TRY_ALLOC(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
        d->own_stream = true;
    }
    for (Stage &g : d->stage) {
        DEV_ALLOC(g.d_bits, d->bits_cap);
        TRY_ALLOC(hipHostMalloc(&g.h_bits, d->bits_cap));
        DEV_ALLOC(g.d_slices, sizeof(SliceDesc) * d->slices_cap);
        TRY_ALLOC(hipHostMalloc(&g.h_slices, sizeof(SliceDesc) * d->slices_cap));
        DEV_ALLOC(g.d_pics, sizeof(PicDesc) * d->pics_cap);
        TRY_ALLOC(hipHostMalloc(&g.h_pics, sizeof(PicDesc) * d->pics_cap));
        DEV_ALLOC(g.d_status, sizeof(uint32_t) * 8 * d->slices_cap);
        TRY_ALLOC(hipHostMalloc(&g.h_status, sizeof(uint32_t) * 8 * d->slices_cap));
        DEV_ALLOC(g.d_lists, sizeof(uint32_t) * 6 * d->pics_cap);
        TRY_ALLOC(hipHostMalloc(&g.h_lists, sizeof(uint32_t) * 6 * d->pics_cap));
        DEV_ALLOC(g.d_bext, sizeof(BSliceExt) * d->slices_cap);
        TRY_ALLOC(hipHostMalloc(&g.h_bext, sizeof(BSliceExt) * d->slices_cap));
        TRY_ALLOC(hipEventCreateWithFlags(&g.ev_upload, hipEventDisableTiming));
        TRY_ALLOC(hipEventCreateWithFlags(&g.ev_done, hipEventDisableTiming));
        g.out.resize(S);
        memset(&g.info, 0, sizeof(g.info));
    }
    {
        // Optional experiment knob: H264MI_ENT_CUS=n gives the entropy streams the first n compute units
        // and reconstruction the rest (hipExtStreamCreateWithCUMask).  Measured slower than sharing the
        // whole chip at every split tried (96..192 of 256), so the default is no partition.
        hipDeviceProp_t prop;
        TRY_ALLOC(hipGetDeviceProperties(&prop, cfg->device));
        const int ncu = prop.multiProcessorCount;
        int ent_cus = 0;
#if defined(H264MI_TEST_HOOKS) /* measurement switches of the hooks build (libh264mi_hooks.so) */
        if (const char *e = getenv("H264MI_ENT_LDS_PAD")) d->ent_lds_pad = static_cast<size_t>(atoi(e));
        if (const char *e = getenv("H264MI_ENT_CUS")) ent_cus = atoi(e);
#endif
        const int words = (ncu + 31) / 32;
        std::vector<uint32_t> me(words, 0), mr(words, 0);
        bool split = ent_cus > 0 && ent_cus < ncu;
        for (int i = 0; i < ncu; i++) {
            // interleave in blocks of 8 so that both partitions span all XCDs / shader engines
            const int k = i / 8;
            bool ent = split ? (((k + 1) * ent_cus / ncu) != (k * ent_cus / ncu)) : true;
            if (ent) me[i / 32] |= 1u << (i % 32);
            if (!ent || !split) mr[i / 32] |= 1u << (i % 32);
        }
        for (int i = 0; i < 2; i++) {
            if (split)
                TRY_ALLOC(hipExtStreamCreateWithCUMask(&d->ent_stream[i], words, me.data()));
            else
                TRY_ALLOC(hipStreamCreateWithFlags(&d->ent_stream[i], hipStreamNonBlocking));
        }
        if (split)
            TRY_ALLOC(hipExtStreamCreateWithCUMask(&d->rec_stream, words, mr.data()));
        else
            TRY_ALLOC(hipStreamCreateWithFlags(&d->rec_stream, hipStreamNonBlocking));
        TRY_ALLOC(hipEventCreateWithFlags(&d->ev_user, hipEventDisableTiming));
    }
    for (int i = 0; i < MI_SETS; i++) {
        DEV_ALLOC(d->d_mbrec[i], sizeof(MbRec) * d->mb_cap);
        DEV_ALLOC(d->d_dbprm[i], sizeof(DbPrm) * d->mb_cap);
        DEV_ALLOC(d->d_imask[i], sizeof(unsigned long long) * (d->mb_cap / 64 + 2));
        if (i == 0) {
            // Pool size.  The worst case is 26 blocks (832 bytes) per macroblock; real streams code a fraction of that (the
            // 1080p QP 28 bench streams: ~5 blocks per macroblock).  Small decoders get the worst case; large ones 8 blocks
            // per macroblock, at least 1 GiB -- a batch that needs more fails with H264MI_EDECODE ("coefficient pool
            // exhausted", code 40) instead of reserving 3 x 52 GB for a case that does not occur.  H264MI_COEF_BLOCKS_PER_MB overrides.
            const uint64_t worst = d->mb_cap * MI_COEF_BLOCKS + static_cast<uint64_t>(d->slices_cap + 1) * MI_COEF_CHUNK;
            uint64_t per_mb = 8;
            if (const char *e = getenv("H264MI_COEF_BLOCKS_PER_MB")) per_mb = static_cast<uint64_t>(std::min(std::max(atoi(e), 1), MI_COEF_BLOCKS));
            if (d->cfg.coef_blocks_per_mb > 0) per_mb = static_cast<uint64_t>(std::min<int>(d->cfg.coef_blocks_per_mb, MI_COEF_BLOCKS));
            // (an explicit coef_blocks_per_mb is taken at its word; the default has a floor of 1 GiB)
            const uint64_t typical = (d->cfg.coef_blocks_per_mb > 0 ? d->mb_cap * per_mb : std::max<uint64_t>(d->mb_cap * per_mb, (1ull << 30) / 32)) +
                                     static_cast<uint64_t>(d->slices_cap + 1) * MI_COEF_CHUNK;
            d->pool_blocks = std::min<uint64_t>(std::min<uint64_t>(worst, typical), 0xFFFF0000ull / MI_SETS);
            DEV_ALLOC(d->d_pool_head, sizeof(uint32_t) * MI_SETS);
        }
        // the MI_SETS pools are ONE allocation: a pass that ran out of residual blocks is repeated on its own with all of it (retry_exhausted)
        if (i == 0) DEV_ALLOC(d->d_coef[0], d->pool_blocks * 32 * MI_SETS);
        else d->d_coef[i] = d->d_coef[0] + static_cast<size_t>(i) * d->pool_blocks * 16;
        DEV_ALLOC(d->d_toprows[i], static_cast<size_t>(d->slices_cap) * (d->Wmax / 16) * MI_TOPROW_BYTES);
        TRY_ALLOC(hipEventCreateWithFlags(&d->ev_ent[i], hipEventDisableTiming));
        TRY_ALLOC(hipEventCreateWithFlags(&d->ev_col[i], hipEventDisableTiming));
        TRY_ALLOC(hipEventCreateWithFlags(&d->ev_rec[i], hipEventDisableTiming));
    }
    DEV_ALLOC(d->d_pools, sizeof(FramePool) * S);
    DEV_ALLOC(d->d_frames, d->slot_bytes * d->n_slots * S + 256); // (K4's unaligned dword loads may read 3 bytes past a plane)
    // 80 bytes per macroblock and frame slot: whatever a later B picture may need of a reference picture's motion (8.4.1.2.1)
    d->colrec_per_slot = static_cast<size_t>(d->Wmax / 16) * (d->Hmax / 16);
    // (the ColRec arrays themselves -- 12.9 GB for 256 streams of 1080p -- are allocated when the first B slice arrives: ensure_b_buffers)
    DEV_ALLOC(d->d_tables, sizeof(DevTables));
    TRY_ALLOC(hipHostMalloc(&d->h_tables, sizeof(DevTables)));
    // K5 keeps a whole macroblock row per in-flight group in dynamic LDS (up to 320 columns): opt in beyond 64 KB
    TRY_ALLOC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_deblock), hipFuncAttributeMaxDynamicSharedMemorySize, MI_DEBLOCK_LDS_MAX));
    TRY_ALLOC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_deblock_x), hipFuncAttributeMaxDynamicSharedMemorySize, MI_DEBLOCK_LDS_MAX));
    // cross-workgroup hand-off state of the banded kernels; H264MI_X_WGS = 0 switches them off, n: up to n workgroups per launch
    if (const char *e = getenv("H264MI_X_WGS")) d->x_max_wgs = std::min(std::max(atoi(e), 0), d->x_cap);
#if defined(H264MI_TEST_HOOKS)
    if (const char *e = getenv("H264MI_K5_WAVES")) d->k5_max_waves = std::min(std::max(atoi(e), 1), MI_DEBLOCK8_MAX_WAVES); // measurement: fewer wavefronts per picture in k_deblock
#endif
    // (test hook: where the launch epoch and the ticket counters start, so that a test can cross their 32-bit wrap)
    DEV_ALLOC(d->d_xring, static_cast<size_t>(d->x_cap) * (d->Wmax / 16) * 24 * sizeof(unsigned long long));
    DEV_ALLOC(d->d_xdone, static_cast<size_t>(d->x_cap3) * (d->Wmax / 16) * sizeof(uint32_t));
    DEV_ALLOC(d->d_xctl, 8 * 128);
    TRY_ALLOC(hipHostMalloc(&d->h_xstatus, sizeof(uint32_t)));
    *d->h_xstatus = 0;
    TRY_ALLOC(hipMemset(d->d_xring, 0, static_cast<size_t>(d->x_cap) * (d->Wmax / 16) * 24 * sizeof(unsigned long long)));
    TRY_ALLOC(hipMemset(d->d_xdone, 0, static_cast<size_t>(d->x_cap3) * (d->Wmax / 16) * sizeof(uint32_t)));
    TRY_ALLOC(hipMemset(d->d_xctl, 0, 8 * 128));
    build_tables(d->h_tables);
    d->h_pools.resize(S);
    for (int si = 0; si < S; si++) { // static per stream (kernels take the geometry of a picture from its PicDesc)
        FramePool &fp = d->h_pools[si];
        fp.base = reinterpret_cast<uint64_t>(d->d_frames) + static_cast<uint64_t>(si) * d->slot_bytes * d->n_slots;
        fp.slot_bytes = d->slot_bytes;
        fp.w = fp.h = 0;
        fp.n_slots = static_cast<uint32_t>(d->n_slots);
        fp.pad = 0;
    }
    TRY_ALLOC(hipMemcpy(d->d_pools, d->h_pools.data(), sizeof(FramePool) * S, hipMemcpyHostToDevice));
    // a deterministic background for macroblocks no slice covers
    TRY_ALLOC(hipMemsetAsync(d->d_frames, 128, d->slot_bytes * d->n_slots * S, d->stream));
    TRY_ALLOC(hipStreamSynchronize(d->stream));
#undef TRY_ALLOC
#undef DEV_ALLOC
    if (d->cfg.b_pictures) { // the caller expects B pictures: their buffers now, motion kept from the first picture on
        r = ensure_b_buffers(d);
        if (r != H264MI_OK) {
            free_all(d);
            delete d;
            return r;
        }
    }
    *out = d;
    return H264MI_OK;
}

extern "C" int32_t h264mi_decoder_destroy(h264mi_decoder *d) {
    if (!d) return H264MI_EINVAL;
    GUARD(d);
    for (int i = 0; i < 2; i++) hipStreamSynchronize(d->ent_stream[i]);
    hipStreamSynchronize(d->rec_stream);
    hipStreamSynchronize(d->stream);
    free_all(d);
    delete d;
    return H264MI_OK;
}
extern "C" int32_t h264mi_decoder_set_stream(h264mi_decoder *d, void *s) {
    if (!d) return H264MI_EINVAL;
    GUARD(d);
    hipStreamSynchronize(d->stream);
    if (d->own_stream) hipStreamDestroy(d->stream);
    d->own_stream = false;
    d->stream = static_cast<hipStream_t>(s);
    return H264MI_OK;
}
// Forget everything about a stream: parameter sets, reference pictures, POC / frame_num history, outputs.
static void reset_stream(StreamState &s, bool keep_parameter_sets) {
    for (auto &sl : s.slots) sl = Slot();
    s.epoch++;
    s.cur_slot = s.cur_pic = -1, s.cur_slices = 0;
    s.cur_field = 0, s.cur_second = false, s.pend_slot = -1;
    s.n_pics_in_batch = 0;
    s.prev_poc_msb = s.prev_poc_lsb = s.prev_frame_num = s.prev_frame_num_offset = s.prev_ref_frame_num = 0;
    if (!keep_parameter_sets) {
        memset(s.sps_ok, 0, sizeof(s.sps_ok));
        memset(s.pps_ok, 0, sizeof(s.pps_ok));
        s.active_sps = -1, s.wmb = s.hmb = 0;
        s.need_idr = false, s.status = H264MI_OK;
    }
}
extern "C" int32_t h264mi_decoder_reset(h264mi_decoder *d) {
    if (!d) return H264MI_EINVAL;
    for (auto &s : d->st) reset_stream(s, true);
    for (Stage &g : d->stage) {
        g.prepared = false;
        for (auto &o : g.out) o.clear();
    }
    return H264MI_OK;
}
extern "C" int32_t h264mi_stream_reset(h264mi_decoder *d, int32_t stream) {
    if (!d || stream < 0 || stream >= static_cast<int>(d->st.size())) return H264MI_EINVAL;
    reset_stream(d->st[stream], false);
    for (Stage &g : d->stage) g.out[stream].clear();
    return H264MI_OK;
}
extern "C" int32_t h264mi_stream_status(h264mi_decoder *d, int32_t stream, int32_t *status) {
    if (!d || !status || stream < 0 || stream >= static_cast<int>(d->st.size())) return H264MI_EINVAL;
    *status = d->st[stream].status;
    return H264MI_OK;
}
extern "C" int32_t h264mi_decoder_set_isolation(h264mi_decoder *d, int32_t on) {
    if (!d) return H264MI_EINVAL;
    d->isolate = on != 0;
    return H264MI_OK;
}
extern "C" int32_t h264mi_decoder_memory(h264mi_decoder *d, int64_t *device_bytes) {
    if (!d || !device_bytes) return H264MI_EINVAL;
    *device_bytes = static_cast<int64_t>(d->dev_bytes);
    return H264MI_OK;
}
extern "C" int32_t h264mi_decoder_unpinned_failures(h264mi_decoder *d, int64_t *n) {
    if (!d || !n) return H264MI_EINVAL;
    GUARD(d);
    *n = d->unpinned_failures;
    return H264MI_OK;
}

extern "C" int32_t h264mi_decoder_coef_pool(h264mi_decoder *d, int64_t *used_blocks, int64_t *capacity_blocks) {
    if (!d || !used_blocks || !capacity_blocks) return H264MI_EINVAL;
    uint32_t heads[MI_SETS] = {0};
    HIP_TRY(hipMemcpy(heads, d->d_pool_head, sizeof(heads), hipMemcpyDeviceToHost));
    uint32_t m = 0;
    for (int i = 0; i < MI_SETS; i++) m = std::max(m, heads[i]);
    *used_blocks = static_cast<int64_t>(m), *capacity_blocks = static_cast<int64_t>(d->pool_blocks);
    return H264MI_OK;
}
extern "C" int32_t h264mi_decoder_set_profiling(h264mi_decoder *d, int32_t on) {
    if (!d) return H264MI_EINVAL;
    d->profiling = on != 0;
    return H264MI_OK;
}

// ---------------------------------------------------------------- picture management (8.2)
static int compute_poc(StreamState &s, const h264mi_sps &sps, const h264mi_slice_header &sh) { // 8.2.1
    const bool idr = sh.nal_unit_type == 5;
    const int max_fn = 1 << (sps.log2_max_frame_num_minus4 + 4);
    int poc = 0;
    if (sps.pic_order_count_type == 0) {
        const int max_lsb = 1 << (sps.log2_max_pic_order_cnt_lsb_min4 + 4);
        int prev_msb = idr ? 0 : s.prev_poc_msb, prev_lsb = idr ? 0 : s.prev_poc_lsb, msb;
        if (sh.pic_order_cnt_lsb < prev_lsb && prev_lsb - sh.pic_order_cnt_lsb >= max_lsb / 2)
            msb = prev_msb + max_lsb;
        else if (sh.pic_order_cnt_lsb > prev_lsb && sh.pic_order_cnt_lsb - prev_lsb > max_lsb / 2)
            msb = prev_msb - max_lsb;
        else
            msb = prev_msb;
        const int top = msb + sh.pic_order_cnt_lsb, bot = top + sh.delta_pic_order_cnt_bottom; // (8-4 / 8-5: a field picture has the one count, delta is 0)
        poc = std::min(top, bot);
        s.top_above_poc = top - poc;
        s.poc_top = top, s.poc_bot = bot;
        if (sh.nal_ref_idc) s.prev_poc_msb = msb, s.prev_poc_lsb = sh.pic_order_cnt_lsb;
    } else {
        int fno = idr ? 0 : (s.prev_frame_num > sh.frame_num ? s.prev_frame_num_offset + max_fn : s.prev_frame_num_offset);
        if (sps.pic_order_count_type == 1) {
            int n = sps.num_ref_frames_in_pic_order_cnt_cycle;
            int abs_fn = n ? fno + sh.frame_num : 0;
            if (!sh.nal_ref_idc && abs_fn > 0) abs_fn--;
            int expected = 0;
            if (abs_fn > 0) {
                int cyc = (abs_fn - 1) / n, in_cyc = (abs_fn - 1) % n, delta = 0;
                for (int i = 0; i < n; i++) delta += sps.offset_for_ref_frame_list[i];
                expected = cyc * delta;
                for (int i = 0; i <= in_cyc; i++) expected += sps.offset_for_ref_frame_list[i];
            }
            if (!sh.nal_ref_idc) expected += sps.offset_for_non_ref_pic;
            const int top = expected + sh.delta_pic_order_cnt[0], bot = top + sps.offset_for_top_to_bottom_field + sh.delta_pic_order_cnt[1];
            if (sh.field_pic) // 8-10: a bottom field is at expected + offset_for_top_to_bottom_field + delta_pic_order_cnt[0]
                poc = sh.bottom_field ? expected + sps.offset_for_top_to_bottom_field + sh.delta_pic_order_cnt[0] : top, s.poc_top = s.poc_bot = poc;
            else
                poc = std::min(top, bot), s.poc_top = top, s.poc_bot = bot;
        } else {
            poc = idr ? 0 : (sh.nal_ref_idc ? 2 * (fno + sh.frame_num) : 2 * (fno + sh.frame_num) - 1);
            s.poc_top = s.poc_bot = poc;
        }
        s.prev_frame_num_offset = fno;
    }
    s.prev_frame_num = sh.frame_num;
    return poc;
}

// 8.2.4 for a field picture (8.2.4.2.2 / 8.2.4.2.4 + 8.2.4.2.5, modification 8.2.4.3 with the field picture numbers of 8.2.4.1): the lists hold
// FIELDS, written as frame slot | parity << 14 (MI_REF_PARITY).  The reference frames are put in order first -- P: by FrameNumWrap, the frame
// of the current field included when this is its second field and the first one is a reference; B: by PicOrderCnt around the current field,
// list 0 the earlier ones nearest first and then the later ones, list 1 the other way round; long-term frames by LongTermFrameIdx --, then
// their fields are taken alternately, the parity of the current field first; a frame that lacks the wanted field is passed over, and when one
// parity is used up the rest of the other one follows in order.
static int build_ref_lists_field(StreamState &s, const h264mi_sps &sps, const h264mi_slice_header &sh, bool bslice, int16_t *out0, int16_t *out1) {
    const int max_fn = 1 << (sps.log2_max_frame_num_minus4 + 4);
    const int bottom = sh.bottom_field ? 1 : 0;
    auto usable = [&](int slot, int par) { return ((s.slots[slot].fields & ~s.slots[slot].funref) >> par) & 1; };
    std::vector<int> st, lt;
    for (int i = 0; i < static_cast<int>(s.slots.size()); i++) {
        Slot &sl = s.slots[i];
        if (i == s.cur_slot && !(s.cur_second && sl.ref == 1)) continue; // (a second field may predict from the first field of its frame)
        if (sl.ref == 1) {
            sl.frame_num_wrap = sl.frame_num > sh.frame_num ? sl.frame_num - max_fn : sl.frame_num;
            st.push_back(i);
        } else if (sl.ref == 2)
            lt.push_back(i);
    }
    std::sort(lt.begin(), lt.end(), [&](int a, int b) { return s.slots[a].long_idx < s.slots[b].long_idx; });
    if (st.empty() && lt.empty()) {
        set_error("P/B slice without reference pictures");
        return H264MI_EBITSTREAM;
    }
    std::vector<int> ord[2];
    if (!bslice) {
        std::sort(st.begin(), st.end(), [&](int a, int b) { return s.slots[a].frame_num_wrap > s.slots[b].frame_num_wrap; });
        ord[0] = st;
    } else {
        // PicOrderCnt of a reference frame here: the smaller of its fields' (Slot::poc); of the current frame (second field): its first field's
        const int cur_poc = s.slots[s.cur_slot].fpoc[bottom];
        std::vector<std::pair<int, int>> before, after; // (PicOrderCnt, slot)
        for (int i : st) {
            if (s.slots[i].nonexisting) continue;
            const int fp = i == s.cur_slot ? s.slots[i].fpoc[!bottom] : s.slots[i].poc;
            (fp <= cur_poc ? before : after).push_back({fp, i});
        }
        std::stable_sort(before.begin(), before.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first > b.first; });
        std::stable_sort(after.begin(), after.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
        for (auto &e : before) ord[0].push_back(e.second);
        for (auto &e : after) ord[0].push_back(e.second), ord[1].push_back(e.second);
        for (auto &e : before) ord[1].push_back(e.second);
    }
    std::vector<int> lists[2];
    for (int l = 0; l < (bslice ? 2 : 1); l++)
        for (int grp = 0; grp < 2; grp++) { // short-term frames, then long-term frames: each group alternates on its own
            const std::vector<int> &fr = grp ? lt : ord[l];
            const int nfr = static_cast<int>(fr.size());
            int a = 0, b = 0; // next frame to look at for the same / the opposite parity
            for (int want_same = 1;; want_same ^= 1) {
                int &cursor = want_same ? a : b;
                const int par = want_same ? bottom : !bottom;
                while (cursor < nfr && !usable(fr[cursor], par)) cursor++;
                if (cursor == nfr) { // this parity is used up: the rest of the other one
                    int &other = want_same ? b : a;
                    for (; other < nfr; other++)
                        if (usable(fr[other], !par)) lists[l].push_back(fr[other] | (!par ? MI_REF_PARITY : 0));
                    break;
                }
                lists[l].push_back(fr[cursor] | (par ? MI_REF_PARITY : 0));
                cursor++;
            }
        }
    if (bslice && lists[1].size() > 1 && lists[1] == lists[0]) std::swap(lists[1][0], lists[1][1]);
    const int max_pic_num = 2 * max_fn, cur_pic_num = 2 * sh.frame_num + 1; // 8.2.4.1: MaxPicNum, CurrPicNum of a field
    for (int l = 0; l < (bslice ? 2 : 1); l++) {
        std::vector<int> &list = lists[l];
        const int nact = (l ? sh.num_ref_idx_l1_active_minus1 : sh.num_ref_idx_l0_active_minus1) + 1;
        if (nact > MI_MAX_REFS) {
            set_error("num_ref_idx_l%d_active %d > %d reference fields is out of scope", l, nact, MI_MAX_REFS);
            return H264MI_EUNSUPPORTED;
        }
        list.resize(nact, -1);
        list.resize(nact + 1, -1);
        const int32_t *idcs = l ? sh.modification_of_pic_nums_l1 : sh.modification_of_pic_nums, *vals = l ? sh.modification_value_l1 : sh.modification_value;
        const int nmod = l ? sh.n_ref_pic_list_modifications_l1 : sh.n_ref_pic_list_modifications;
        if (l ? sh.ref_pic_list_modification_flag_l1 : sh.ref_pic_list_modification_flag_l0) { // 8.2.4.3 on field picture numbers
            int pred = cur_pic_num, idx = 0;
            for (int k = 0; k < nmod && idx < nact; k++) {
                int target = -1;
                if (idcs[k] < 2) {
                    const int diff = vals[k] + 1;
                    if (idcs[k] == 0) {
                        pred -= diff;
                        if (pred < 0) pred += max_pic_num;
                    } else {
                        pred += diff;
                        if (pred >= max_pic_num) pred -= max_pic_num;
                    }
                    const int picnum = pred > cur_pic_num ? pred - max_pic_num : pred;
                    for (int i : st)
                        for (int par = 0; par < 2; par++) // picNumF: 2 * FrameNumWrap + 1 for a field of the current parity, 2 * FrameNumWrap for the other
                            if (usable(i, par) && 2 * s.slots[i].frame_num_wrap + (par == bottom) == picnum) target = i | (par ? MI_REF_PARITY : 0);
                } else
                    for (int i : lt)
                        for (int par = 0; par < 2; par++)
                            if (usable(i, par) && 2 * s.slots[i].long_idx + (par == bottom) == vals[k]) target = i | (par ? MI_REF_PARITY : 0);
                if (target < 0) {
                    set_error("ref_pic_list_modification names a missing field");
                    return H264MI_EBITSTREAM;
                }
                for (int c = nact; c > idx; c--) list[c] = list[c - 1];
                list[idx++] = target;
                int nidx = idx;
                for (int c = idx; c <= nact; c++)
                    if (list[c] != target) list[nidx++] = list[c];
            }
        }
        int16_t *out = l ? out1 : out0;
        for (int i = 0; i < MI_MAX_REFS; i++) out[i] = static_cast<int16_t>(i < nact ? list[i] : -1);
    }
    return H264MI_OK;
}

// 8.2.4: RefPicList0 (P and B slices) and RefPicList1 (B slices) as frame-pool slots
static int build_ref_lists(StreamState &s, const h264mi_sps &sps, const h264mi_slice_header &sh, bool bslice, int16_t *out0 /*MI_MAX_REFS*/, int16_t *out1) {
    if (sh.field_pic) return build_ref_lists_field(s, sps, sh, bslice, out0, out1);
    const int max_fn = 1 << (sps.log2_max_frame_num_minus4 + 4);
    std::vector<int> st, lt;
    for (int i = 0; i < static_cast<int>(s.slots.size()); i++) {
        Slot &sl = s.slots[i];
        if (i == s.cur_slot) continue;
        // 8.2.4.2.1: a frame picture predicts from frames (or complementary field pairs) of which BOTH fields are reference fields
        if (sl.ref && (sl.fields != 3 || sl.funref)) continue;
        if (sl.ref == 1) {
            sl.frame_num_wrap = sl.frame_num > sh.frame_num ? sl.frame_num - max_fn : sl.frame_num;
            sl.pic_num = sl.frame_num_wrap;
            st.push_back(i);
        } else if (sl.ref == 2) {
            sl.pic_num = sl.long_idx;
            lt.push_back(i);
        }
    }
    std::sort(lt.begin(), lt.end(), [&](int a, int b) { return s.slots[a].long_idx < s.slots[b].long_idx; });
    if (st.empty() && lt.empty()) {
        set_error("P/B slice without reference pictures");
        return H264MI_EBITSTREAM;
    }
    std::vector<int> lists[2];
    if (!bslice) { // 8.2.4.2.1: PicNum descending, then LongTermPicNum ascending
        std::sort(st.begin(), st.end(), [&](int a, int b) { return s.slots[a].pic_num > s.slots[b].pic_num; });
        lists[0] = st;
    } else { // 8.2.4.2.3: by PicOrderCnt relative to the current picture
        const int cur_poc = s.slots[s.cur_slot].poc;
        std::vector<int> before, after;
        for (int i : st) {
            if (s.slots[i].nonexisting && sps.pic_order_count_type == 0) continue; // 8.2.4.2.3: no PicOrderCnt, not in the lists of B slices
            (s.slots[i].poc < cur_poc ? before : after).push_back(i);
        }
        std::sort(before.begin(), before.end(), [&](int a, int b) { return s.slots[a].poc > s.slots[b].poc; });
        std::sort(after.begin(), after.end(), [&](int a, int b) { return s.slots[a].poc < s.slots[b].poc; });
        lists[0] = before;
        lists[0].insert(lists[0].end(), after.begin(), after.end());
        lists[1] = after;
        lists[1].insert(lists[1].end(), before.begin(), before.end());
        lists[1].insert(lists[1].end(), lt.begin(), lt.end());
    }
    lists[0].insert(lists[0].end(), lt.begin(), lt.end());
    if (bslice && lists[1].size() > 1 && lists[1] == lists[0]) std::swap(lists[1][0], lists[1][1]);
    for (int l = 0; l < (bslice ? 2 : 1); l++) {
        std::vector<int> &list = lists[l];
        const int nact = (l ? sh.num_ref_idx_l1_active_minus1 : sh.num_ref_idx_l0_active_minus1) + 1;
        if (nact > MI_MAX_REFS) {
            set_error("num_ref_idx_l%d_active %d > %d (field refs are out of scope)", l, nact, MI_MAX_REFS);
            return H264MI_EUNSUPPORTED;
        }
        list.resize(nact, -1); // the initial list is cut (or padded with "no reference picture") to the active size
        list.resize(nact + 1, -1);
        const int32_t *idcs = l ? sh.modification_of_pic_nums_l1 : sh.modification_of_pic_nums, *vals = l ? sh.modification_value_l1 : sh.modification_value;
        const int nmod = l ? sh.n_ref_pic_list_modifications_l1 : sh.n_ref_pic_list_modifications;
        if (l ? sh.ref_pic_list_modification_flag_l1 : sh.ref_pic_list_modification_flag_l0) { // 8.2.4.3
            int pred = sh.frame_num, idx = 0;
            for (int k = 0; k < nmod && idx < nact; k++) {
                int target = -1;
                if (idcs[k] < 2) {
                    int diff = vals[k] + 1;
                    if (idcs[k] == 0) {
                        pred -= diff;
                        if (pred < 0) pred += max_fn;
                    } else {
                        pred += diff;
                        if (pred >= max_fn) pred -= max_fn;
                    }
                    int picnum = pred > sh.frame_num ? pred - max_fn : pred;
                    for (int i : st)
                        if (s.slots[i].pic_num == picnum) target = i;
                } else
                    for (int i : lt)
                        if (s.slots[i].long_idx == vals[k]) target = i;
                if (target < 0) {
                    set_error("ref_pic_list_modification names a missing picture");
                    return H264MI_EBITSTREAM;
                }
                for (int c = nact; c > idx; c--) list[c] = list[c - 1];
                list[idx++] = target;
                int nidx = idx;
                for (int c = idx; c <= nact; c++)
                    if (list[c] != target) list[nidx++] = list[c];
            }
        }
        int16_t *out = l ? out1 : out0;
        for (int i = 0; i < MI_MAX_REFS; i++) out[i] = static_cast<int16_t>(i < nact ? list[i] : -1);
    }
    return H264MI_OK;
}

// 8.2.5: marking after the current picture is complete
static void mark_reference(StreamState &s, const h264mi_sps &sps) {
    const h264mi_slice_header &sh = s.first_sh;
    Slot &cur = s.slots[s.cur_slot];
    const int max_fn = 1 << (sps.log2_max_frame_num_minus4 + 4);
    if (sh.nal_ref_idc) s.prev_ref_frame_num = sh.frame_num; // (operation 5 below: 0)
    if (s.cur_field && sh.nal_ref_idc && sh.nal_unit_type != 5 && sh.adaptive_ref_pic_marking_mode_flag) {
        // 8.2.5.4.1 in a field picture: picNumX names a FIELD (8.2.4.1); the frame stays in the window while its other field is a reference.
        // (operations 2..6 on fields were refused when the picture started)
        const int bottom = s.cur_field == 2, cur_pic_num = 2 * sh.frame_num + 1;
        for (int k = 0; k < sh.n_memory_management_control_operations; k++) {
            const int picnum = cur_pic_num - (sh.mmco_arg1[k] + 1);
            for (auto &sl : s.slots) {
                if (sl.ref != 1) continue;
                const int wrap = sl.frame_num > sh.frame_num ? sl.frame_num - max_fn : sl.frame_num;
                for (int par = 0; par < 2; par++)
                    if ((((sl.fields & ~sl.funref) >> par) & 1) && 2 * wrap + (par == bottom) == picnum) {
                        sl.funref |= 1 << par;
                        if (!(sl.fields & ~sl.funref)) sl.ref = 0; // (the current frame, whose other field is being decoded, is marked just below)
                    }
            }
        }
        cur.ref = 1;
        return;
    }
    // 8.2.5.3: the second field of a frame whose first field is a reference joins it; nothing leaves the window
    if (s.cur_field && s.cur_second && cur.ref) return;
    if (!sh.nal_ref_idc) {
        cur.ref = 0;
        return;
    }
    if (sh.nal_unit_type == 5) {
        for (auto &sl : s.slots) sl.ref = 0;
        cur.ref = sh.long_term_reference_flag ? 2 : 1;
        cur.long_idx = 0;
        return;
    }
    cur.ref = 1;
    if (sh.adaptive_ref_pic_marking_mode_flag) {
        for (int k = 0; k < sh.n_memory_management_control_operations; k++) {
            int op = sh.memory_management_control_operation[k];
            for (auto &sl : s.slots)
                if (sl.ref == 1) sl.pic_num = sl.frame_num > sh.frame_num ? sl.frame_num - max_fn : sl.frame_num;
            if (op == 1 || op == 3) {
                int picnum = sh.frame_num - (sh.mmco_arg1[k] + 1);
                for (auto &sl : s.slots)
                    if (&sl != &cur && sl.ref == 1 && sl.pic_num == picnum) {
                        if (op == 1)
                            sl.ref = 0;
                        else {
                            for (auto &o : s.slots)
                                if (o.ref == 2 && o.long_idx == sh.mmco_arg2[k]) o.ref = 0;
                            sl.ref = 2, sl.long_idx = sh.mmco_arg2[k];
                        }
                    }
            } else if (op == 2) {
                for (auto &sl : s.slots)
                    if (sl.ref == 2 && sl.long_idx == sh.mmco_arg1[k]) sl.ref = 0;
            } else if (op == 4) {
                for (auto &sl : s.slots)
                    if (sl.ref == 2 && sl.long_idx >= sh.mmco_arg1[k]) sl.ref = 0;
            } else if (op == 5) {
                for (auto &sl : s.slots)
                    if (&sl != &cur) sl.ref = 0;
                cur.frame_num = 0, cur.poc = 0; // 8.2.1: tempPicOrderCnt is subtracted, the picture ends up at PicOrderCnt 0
                {
                    const int m = std::min(cur.fpoc[0], cur.fpoc[1]);
                    cur.fpoc[0] -= m, cur.fpoc[1] -= m;
                }
                s.prev_frame_num = s.prev_frame_num_offset = s.prev_poc_msb = s.prev_ref_frame_num = 0;
                // 8.2.1.1: prevPicOrderCntLsb = TopFieldOrderCnt after tempPicOrderCnt was subtracted -- 0 unless the bottom field is the earlier one
                s.prev_poc_lsb = sps.pic_order_count_type == 0 ? s.top_above_poc : 0;
            } else if (op == 6) {
                for (auto &o : s.slots)
                    if (o.ref == 2 && o.long_idx == sh.mmco_arg2[k]) o.ref = 0;
                cur.ref = 2, cur.long_idx = sh.mmco_arg2[k];
            }
        }
    } else { // sliding window 8.2.5.3
        int nref = 0, maxref = std::max(sps.max_num_ref_frames, 1);
        Slot *oldest = nullptr;
        for (auto &sl : s.slots) {
            if (&sl == &cur || !sl.ref) continue;
            nref++;
            if (sl.ref == 1) {
                sl.frame_num_wrap = sl.frame_num > sh.frame_num ? sl.frame_num - max_fn : sl.frame_num;
                if (!oldest || sl.frame_num_wrap < oldest->frame_num_wrap) oldest = &sl;
            }
        }
        if (nref >= maxref && oldest) oldest->ref = 0;
    }
}

static void finish_picture(h264mi_decoder *d, int si) {
    StreamState &s = d->st[si];
    Stage &g = d->stage[d->prep];
    if (s.cur_slot < 0) return;
    mark_reference(s, s.sps[s.active_sps]);
    PicDesc &pd = g.h_pics[s.cur_pic];
    pd.n_slices = static_cast<uint32_t>(s.cur_slices);
    if (s.cur_field) {
        // a field: the frame goes out when its second field is complete -- or, if that never comes, when the next picture starts (flush_pending_field)
        Slot &cur = s.slots[s.cur_slot];
        cur.fields |= 1 << (s.cur_field - 1);
        if (cur.fields == 3) {
            cur.poc = std::min(cur.fpoc[0], cur.fpoc[1]);
            s.pend_out.poc = cur.poc;
            if (s.cur_second) s.pend_out.pic2 = s.cur_pic;
            g.out[si].push_back(s.pend_out);
            s.pend_slot = -1;
        } else {
            cur.poc = cur.fpoc[s.cur_field - 1];
            s.pend_slot = s.cur_slot;
        }
    } else if (!g.out[si].empty())
        g.out[si].back().poc = s.slots[s.cur_slot].poc; // operation 5 rewrites it
    // Every macroblock of the picture belongs to exactly one slice wavefront (SliceDesc::fill_from / end_mb): order the
    // slices by first_mb (arbitrary slice order is legal in Baseline); a slice's range ends where the next one starts.
    std::vector<uint32_t> idx(pd.n_slices);
    for (uint32_t i = 0; i < pd.n_slices; i++) idx[i] = pd.first_slice + i;
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return g.h_slices[a].first_mb < g.h_slices[b].first_mb; });
    const uint32_t total = pd.wmb * pd.hmb;
    if (pd.fmo) {
        // slice groups: a slice ends where the next slice of ITS group starts; nothing is filled by the wavefronts (the records of
        // the whole picture are zeroed before the launch: launch_entropy)
        const uint8_t *map = g.h_bits + pd.sgmap_off;
        for (uint32_t i = 0; i < pd.n_slices; i++) {
            SliceDesc &sd = g.h_slices[idx[i]];
            sd.fill_from = sd.first_mb, sd.end_mb = total;
            for (uint32_t j = i + 1; j < pd.n_slices; j++)
                if (map[g.h_slices[idx[j]].first_mb] == map[sd.first_mb]) {
                    sd.end_mb = std::max(g.h_slices[idx[j]].first_mb, sd.first_mb);
                    break;
                }
        }
        g.fmo_pics.push_back(static_cast<uint32_t>(s.cur_pic));
    } else
        for (uint32_t i = 0; i < pd.n_slices; i++) {
            SliceDesc &sd = g.h_slices[idx[i]];
            sd.fill_from = i == 0 ? 0 : sd.first_mb;
            sd.end_mb = i + 1 < pd.n_slices ? std::max(g.h_slices[idx[i + 1]].first_mb, sd.first_mb) : total;
        }
    s.cur_slot = s.cur_pic = -1;
}

// A first field whose second field did not come: the frame goes out with one field decoded, the rows of the other parity painted mid-grey
// (at the start of the batch's reconstruction: whatever the slot held before must not show)
static void flush_pending_field(h264mi_decoder *d, int si) {
    StreamState &s = d->st[si];
    if (s.pend_slot < 0) return;
    Stage &g = d->stage[d->prep];
    const Slot &f = s.slots[s.pend_slot];
    OutFrame o = s.pend_out;
    o.poc = f.poc;
    g.out[si].push_back(o);
    g.grey.push_back({static_cast<uint32_t>(si), static_cast<uint32_t>(s.pend_slot), f.fields == 1 ? 1u : 0u, static_cast<uint32_t>(o.wmb * 16), static_cast<uint32_t>(o.hmb * 16)});
    s.pend_slot = -1;
}

static bool new_picture(const h264mi_sps &sps, const h264mi_slice_header &a, const h264mi_slice_header &b) { // 7.4.1.2.4
    if (a.frame_num != b.frame_num || a.pps_id != b.pps_id) return true;
    if (a.field_pic != b.field_pic || a.bottom_field != b.bottom_field) return true; // (the two fields of a frame are two pictures)
    if ((a.nal_ref_idc == 0) != (b.nal_ref_idc == 0)) return true;
    if ((a.nal_unit_type == 5) != (b.nal_unit_type == 5)) return true;
    if (a.nal_unit_type == 5 && a.idr_pic_id != b.idr_pic_id) return true;
    if (sps.pic_order_count_type == 0 && (a.pic_order_cnt_lsb != b.pic_order_cnt_lsb || a.delta_pic_order_cnt_bottom != b.delta_pic_order_cnt_bottom)) return true;
    if (sps.pic_order_count_type == 1 && (a.delta_pic_order_cnt[0] != b.delta_pic_order_cnt[0] || a.delta_pic_order_cnt[1] != b.delta_pic_order_cnt[1])) return true;
    // not in the list of 7.4.1.2.4, but a consequence of 7.4.3: all slices of a picture carry the same slice_group_change_cycle (the map is the
    // picture's), the same marking script and the same long_term_reference_flag -- a difference means another picture even when frame_num and the
    // picture order count agree (they do after memory management operation 5 resets both)
    if (a.slice_group_change_cycle != b.slice_group_change_cycle) return true;
    if (a.adaptive_ref_pic_marking_mode_flag != b.adaptive_ref_pic_marking_mode_flag || a.n_memory_management_control_operations != b.n_memory_management_control_operations) return true;
    for (int k = 0; k < a.n_memory_management_control_operations; k++)
        if (a.memory_management_control_operation[k] != b.memory_management_control_operation[k] || a.mmco_arg1[k] != b.mmco_arg1[k] || a.mmco_arg2[k] != b.mmco_arg2[k]) return true;
    return false;
}

extern "C" int32_t h264mi_slice_starts_picture(const h264mi_sps *sps, const h264mi_slice_header *prev, const h264mi_slice_header *cur) {
    if (!sps || !prev || !cur) return H264MI_EINVAL;
    return new_picture(*sps, *prev, *cur) ? 1 : 0;
}

static int scaling_set_for(h264mi_decoder *d, const h264mi_pps &p) {
    ScalingSet tmp;
    build_scaling(p.scaling_list_4x4, p.scaling_list_8x8, &tmp);
    for (int i = 0; i < d->n_scaling; i++)
        if (!memcmp(&d->h_tables->scaling[i], &tmp, sizeof(tmp))) return i;
    if (d->n_scaling >= MI_MAX_SCALING_SETS) return -1;
    d->h_tables->scaling[d->n_scaling] = tmp;
    d->tables_dirty = true;
    return d->n_scaling++;
}

// 8.2.5.2 decoding process for gaps in frame_num (h264/sps.go:311-312 parses the flag, nothing in the reference uses it): a
// non-IDR picture whose frame_num is neither PrevRefFrameNum nor its successor says that reference frames are missing.  With
// gaps_in_frame_num_value_allowed_flag every skipped value becomes a "non-existing" short-term frame that goes through the
// sliding window like a decoded one -- it pushes older frames out and takes its place in the initial lists, so that the
// indices of the surviving pictures come out as the encoder meant them.  Without the flag pictures were lost: the stream is
// refused (H264MI_EBITSTREAM; with isolation it alone leaves the batch and waits for its next IDR picture) rather than
// predicted from the wrong pictures.
static int fill_frame_num_gap(h264mi_decoder *d, int si, const h264mi_sps &sps, const h264mi_slice_header &sh) {
    StreamState &s = d->st[si];
    if (sh.nal_unit_type == 5) return H264MI_OK;
    const int max_fn = 1 << (sps.log2_max_frame_num_minus4 + 4), cur_fn = sh.frame_num;
    const int expect = (s.prev_ref_frame_num + 1) % max_fn;
    if (cur_fn == s.prev_ref_frame_num || cur_fn == expect) return H264MI_OK;
    if (!sps.gaps_in_frame_num_value_allowed) {
        set_error("stream %d: frame_num %d after %d: reference pictures are missing", si, cur_fn, s.prev_ref_frame_num);
        return H264MI_EBITSTREAM;
    }
    const int maxref = std::max(sps.max_num_ref_frames, 1);
    for (int fn = expect; fn != cur_fn; fn = (fn + 1) % max_fn) {
        int nref = 0;
        Slot *oldest = nullptr, *slot = nullptr;
        for (auto &sl : s.slots) { // 8.2.5.3 with this frame as the current one
            if (!sl.ref) continue;
            nref++;
            if (sl.ref == 1) {
                sl.frame_num_wrap = sl.frame_num > fn ? sl.frame_num - max_fn : sl.frame_num;
                if (!oldest || sl.frame_num_wrap < oldest->frame_num_wrap) oldest = &sl;
            }
        }
        if (nref >= maxref && oldest) oldest->ref = 0;
        for (auto &sl : s.slots)
            if (!sl.ref && !sl.held && !slot) slot = &sl;
        if (!slot) {
            set_error("stream %d: frame pool exhausted", si);
            return H264MI_ECAPACITY;
        }
        *slot = Slot();
        slot->ref = 1, slot->nonexisting = true, slot->frame_num = fn, slot->fields = 3;
        if (sps.pic_order_count_type != 0) { // 8.2.1: as a reference frame with this frame_num (keeps FrameNumOffset right across a wrap)
            h264mi_slice_header f;
            memset(&f, 0, sizeof(f));
            f.frame_num = fn, f.nal_ref_idc = 1, f.nal_unit_type = 1;
            slot->poc = compute_poc(s, sps, f);
        }
        s.prev_ref_frame_num = fn;
    }
    return H264MI_OK;
}

// What only B pictures need exists once the first B slice has been seen: the list-1 vector arrays (64 bytes per macroblock and
// buffer set) and the ColRec arrays (80 bytes per macroblock and frame slot: the motion direct prediction reads).  Reference
// pictures decoded BEFORE that moment have left no ColRec; the ones of the batch before this one -- where RefPicList1[0] of a
// stream's first B picture can still come from -- are filled in now from that batch's records, which are still resident
// (MI_STAGES staging sets, MI_SETS record sets).  Older ones cannot be: direct prediction from them sees an intra picture.
static int ensure_b_buffers(h264mi_decoder *d) {
    for (int i = 0; i < MI_SETS; i++)
        if (!d->d_mv1[i]) {
            hipError_t e = hipMalloc(&d->d_mv1[i], sizeof(MbMv1) * d->mb_cap);
            if (e != hipSuccess) {
                set_error("hipMalloc of the list-1 vector array failed: %s", hipGetErrorString(e));
                return e == hipErrorOutOfMemory ? H264MI_ENOMEM : H264MI_EDEVICE;
            }
            d->dev_bytes += sizeof(MbMv1) * d->mb_cap;
        }
    if (d->d_colrec) return H264MI_OK;
    // allocate, initialise, and only then publish: a failure half-way must not leave an array behind that the next call takes for ready
    const size_t S = d->st.size(), bytes = sizeof(ColRec) * d->colrec_per_slot * d->n_slots * S;
    ColRec *colrec = nullptr;
    uint32_t *backfill = nullptr;
    hipError_t e = hipMalloc(&colrec, bytes);
    if (e == hipSuccess) e = hipMalloc(&backfill, sizeof(uint32_t) * d->pics_cap);
    hipStream_t up = d->ent_stream[d->pass & 1];
    if (e == hipSuccess) e = hipMemsetAsync(colrec, 0xFF, bytes, up); // refslot / ref -1 everywhere: "intra" (never read: Slot::col_valid guards every use)
    if (e != hipSuccess) {
        if (colrec) hipFree(colrec);
        if (backfill) hipFree(backfill);
        set_error("setting up the co-located motion arrays failed: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? H264MI_ENOMEM : H264MI_EDEVICE;
    }
    d->d_colrec = colrec, d->d_backfill = backfill;
    d->dev_bytes += bytes + sizeof(uint32_t) * d->pics_cap;
    // The reference pictures of the batch executed last -- where RefPicList1[0] of a stream's first B picture can still come from -- get their
    // ColRec arrays now, from that batch's records: only if that batch is the one prepared before this one, its record set is known, and it has
    // FINISHED (its descriptor table is rewritten here; a pass still in flight would read it).  Anything older stays without (col_valid).
    Stage &pv = d->stage[(d->prep + MI_STAGES - 1) % MI_STAGES];
    if (MI_STAGES > 1 && pv.executed && d->pass > 0 && d->last_exec_stage == (d->prep + MI_STAGES - 1) % MI_STAGES) {
        HIP_TRY(hipEventSynchronize(pv.ev_done)); // once per decoder
        std::vector<uint32_t> list;
        auto want = [&](size_t si, const OutFrame &o, int pic, int par) {
            if (pic < 0 || pic >= pv.n_pics) return;
            Slot &sl = d->st[si].slots[o.slot];
            pv.h_pics[pic].save_col = 1;
            pv.h_pics[pic].col_out = reinterpret_cast<uint64_t>(d->d_colrec + (si * d->n_slots + o.slot) * d->colrec_per_slot + (par ? d->colrec_per_slot / 2 : 0));
            sl.col_valid[par] = true;
            list.push_back(static_cast<uint32_t>(pic));
        };
        for (size_t si = 0; si < S; si++)
            for (const OutFrame &o : pv.out[si])
                if (d->st[si].slots[o.slot].ref) { // still a reference picture: a B picture may point at it
                    for (int pic : {o.pic, o.pic2})
                        if (pic >= 0 && pic < pv.n_pics) want(si, o, pic, pv.h_pics[pic].field == 2 ? 1 : 0);
                }
        if (!list.empty()) {
            const int pset = d->last_exec_set;
            HIP_TRY(hipMemcpyAsync(pv.d_pics, pv.h_pics, sizeof(PicDesc) * pv.n_pics, hipMemcpyHostToDevice, up));
            HIP_TRY(hipMemcpyAsync(d->d_backfill, list.data(), sizeof(uint32_t) * list.size(), hipMemcpyHostToDevice, up));
            HIP_TRY(hipStreamSynchronize(up)); // (`list` is pageable host memory; once per decoder)
            hipLaunchKernelGGL(k_dbprep, dim3((pv.mbs_max + MI_DBPREP_MBS - 1) / MI_DBPREP_MBS, (static_cast<uint32_t>(list.size()) + 7u) & ~7u), dim3(256), 0, up, d->d_backfill,
                               pv.d_pics, d->d_tables, d->d_mbrec[pset], d->d_mv1[pset], d->d_dbprm[pset], 1, d->d_imask[pset], static_cast<int>(list.size()));
        }
    }
    return H264MI_OK;
}

// one slice NAL of stream `si`
// `off` / `rlen`: where batch_prepare's parallel pass put the slice's RBSP in the pinned staging buffer (16-byte aligned)
static int add_slice(h264mi_decoder *d, int si, size_t off, size_t rlen, int ref_idc, int type) {
    StreamState &s = d->st[si];
    Stage &g = d->stage[d->prep];
    if (g.n_slices >= d->slices_cap) {
        set_error("more than %d slices in the batch", d->slices_cap);
        return H264MI_ECAPACITY;
    }
    uint8_t *rbsp = g.h_bits + off;
    // peek pps id: first_mb_in_slice, slice_type, pic_parameter_set_id
    BitReader br(rbsp, rlen);
    br.ue();
    br.ue();
    uint32_t pps_id = br.ue();
    if (pps_id > 255 || !s.pps_ok[pps_id]) {
        set_error("stream %d: slice refers to missing PPS %u", si, pps_id);
        return H264MI_EBITSTREAM;
    }
    const h264mi_pps &pps = s.pps[pps_id];
    if (!s.sps_ok[pps.sps_id]) {
        set_error("stream %d: PPS %u refers to missing SPS %d", si, pps_id, pps.sps_id);
        return H264MI_EBITSTREAM;
    }
    const h264mi_sps &sps = s.sps[pps.sps_id];
    h264mi_slice_header sh;
    int r = parse_slice_header(&sps, &pps, ref_idc, type, rbsp, rlen, &sh);
    if (r != H264MI_OK) return r;
    if (sh.redundant_pic_cnt > 0) return H264MI_OK; // redundant pictures are dropped
    if (s.need_idr) { // after an error nothing can be trusted before the next IDR picture
        if (type != 5) return H264MI_OK;
        s.need_idr = false;
    }
    const int st = sh.slice_type % 5;
    if (st > 2) {
        set_error("stream %d: slice_type %d is out of scope (SP / SI slices)", si, sh.slice_type);
        return H264MI_EUNSUPPORTED;
    }
    // chroma_format_idc 0 (monochrome; h264/sps.go:226-243 ChromaFormat, h264/slice.go:179-219 SubWidthC / SubHeightC): the entropy kernels leave out the
    // chroma syntax (PicDesc::mono), and nothing else changes -- the frame pool starts out at 128, intra chroma prediction is the DC of planes that are
    // 128 everywhere, inter prediction copies 128, the residual is zero and the filters leave constants alone: the chroma planes a 4:2:0 display expects
    if (sps.chroma_format > 1 || sps.bit_depth_luma_minus8 || sps.bit_depth_chroma_minus8 || sps.qprime_y_zero_transform_bypass) {
        set_error("stream %d: only 4:2:0 and monochrome 8-bit streams are supported (chroma_format_idc %d; 4:2:2 / 4:4:4 and more than 8 bits are out of scope)", si, sps.chroma_format);
        return H264MI_EUNSUPPORTED;
    }
    // frame_mbs_only_flag = 0 (h264/sps.go:316-322): the pictures are frames (decoded like progressive ones: map units are two macroblock rows
    // high, crop units double) or field pictures (h264/slice.go:867-872: field_pic_flag / bottom_field_flag; PAFF), in any mix.
    // Macroblock-adaptive frame/field coding (MBAFF, h264/slice.go:563-568, 624-634) is not implemented.
    if (!sps.frame_mbs_only && sps.mb_adaptive_frame_field) {
        set_error("stream %d: macroblock-adaptive frame/field coding (MBAFF) is out of scope", si);
        return H264MI_EUNSUPPORTED;
    }
    if (sh.field_pic) {
        if (pps.entropy_coding_mode && !d->cfg.allow_unpinned_field_cabac) {
            // ctxIdx 277..398 and 436..459 (significant_coeff_flag / last_significant_coeff_flag of field-coded blocks, Tables 9-19 .. 9-24): the values in
            // this library's tables were written down without the standard at hand and nothing on its build machine pins them (mi_cabac_mn.cpp) -- a
            // decoder with wrong tables produces plausible, wrong pictures, so it takes an explicit request (h264mi_config.allow_unpinned_field_cabac)
            set_error("stream %d: field pictures with CABAC are refused: the context initialisation values of field-coded blocks (ctxIdx 277-398, 436-459) in "
                      "this library's tables are unpinned (h264mi_config.allow_unpinned_field_cabac = 1 decodes with them); CAVLC field pictures are decoded", si);
            return H264MI_EUNSUPPORTED;
        }
        if (ref_idc && type != 5 && sh.adaptive_ref_pic_marking_mode_flag)
            for (int k = 0; k < sh.n_memory_management_control_operations; k++)
                if (sh.memory_management_control_operation[k] != 1) {
                    set_error("stream %d: memory_management_control_operation %d in a field picture is out of scope (operation 1 is implemented)", si,
                              sh.memory_management_control_operation[k]);
                    return H264MI_EUNSUPPORTED;
                }
    }
    if (sps.max_num_ref_frames > d->cfg.max_ref_frames) {
        set_error("stream %d: max_num_ref_frames %d exceeds the configured max_ref_frames %d", si, sps.max_num_ref_frames, d->cfg.max_ref_frames);
        return H264MI_ECAPACITY;
    }
    const int wmb = sps.pic_width_in_mbs, hmb = sps.pic_height_in_mbs; // of the frame
    const int hmb_pic = sh.field_pic ? hmb / 2 : hmb;                   // of this picture (h264/slice.go:159-176 PicHeightInMbs)
    if (wmb * 16 > d->Wmax || hmb * 16 > d->Hmax || hmb > 320) {
        set_error("stream %d: %dx%d exceeds the configured maximum %dx%d", si, wmb * 16, hmb * 16, d->Wmax, d->Hmax);
        return H264MI_ECAPACITY;
    }
    // (7.4.1.2.4 cannot tell two pictures apart whose headers agree -- e.g. POC type 2 and the frame_num 1 that follows a memory
    // management operation 5 in a picture with frame_num 1 --: a slice that starts where a slice of the current picture already
    // started begins a new picture whatever the headers say.  Not "first_mb_in_slice == 0": with slice groups or arbitrary slice
    // order that slice may come late.)
    const bool restarts = std::find(s.cur_first_mbs.begin(), s.cur_first_mbs.end(), sh.first_mb_in_slice) != s.cur_first_mbs.end();
    if (s.cur_slot >= 0 && (restarts || new_picture(sps, s.first_sh, sh))) finish_picture(d, si);
    if (s.active_sps != pps.sps_id || s.wmb != wmb || s.hmb != hmb) { // (re)activate: new sequence geometry
        if (sh.nal_unit_type != 5 && s.active_sps >= 0 && (s.wmb != wmb || s.hmb != hmb)) {
            set_error("stream %d: picture size changes without an IDR", si);
            return H264MI_EBITSTREAM;
        }
        flush_pending_field(d, si); // (a lone first field of the old sequence goes out before anything of the new one)
        s.active_sps = pps.sps_id, s.wmb = wmb, s.hmb = hmb;
    }
    if (s.cur_slot < 0) { // first slice of a new picture
        // the second field of the frame whose first field was the previous picture (7.4.1.2.4, 3.30 / 3.31): opposite parity, same frame_num, not
        // an IDR picture, reference or not like the first one
        bool second = false;
        if (s.pend_slot >= 0) {
            const Slot &f = s.slots[s.pend_slot];
            if (sh.field_pic && type != 5 && f.fields == (sh.bottom_field ? 1 : 2) && f.frame_num == sh.frame_num && (f.ref != 0) == (ref_idc != 0))
                second = true;
            else
                flush_pending_field(d, si);
        }
        if (!second) {
            r = fill_frame_num_gap(d, si, sps, sh);
            if (r != H264MI_OK) return r;
        }
        // (a field counts as a picture of its own against max_frames_per_batch: include/h264mi.h)
        if (s.n_pics_in_batch >= d->cfg.max_frames_per_batch || g.n_pics >= d->pics_cap) {
            set_error("stream %d: more than %d pictures in one batch", si, d->cfg.max_frames_per_batch);
            return H264MI_ECAPACITY;
        }
        int slot = second ? s.pend_slot : -1;
        for (int i = 0; i < static_cast<int>(s.slots.size()) && slot < 0; i++)
            if (!s.slots[i].ref && !s.slots[i].held) slot = i;
        if (slot < 0) {
            set_error("stream %d: frame pool exhausted", si);
            return H264MI_ECAPACITY;
        }
        if (g.mb_used + static_cast<uint64_t>(wmb) * hmb_pic > d->mb_cap) {
            set_error("macroblock record pool exhausted");
            return H264MI_ECAPACITY;
        }
        s.cur_slot = slot;
        s.cur_pic = g.n_pics++;
        s.cur_slices = 0;
        s.cur_first_mbs.clear();
        s.first_sh = sh;
        s.cur_field = sh.field_pic ? 1 + (sh.bottom_field ? 1 : 0) : 0;
        s.cur_second = second;
        if (second) s.pend_slot = -1; // (it is the current picture's frame now; back in pend_slot only if it still lacks a field when this picture ends)
        Slot &sl = s.slots[slot];
        if (!second) {
            sl = Slot();
            sl.held = true;
            sl.frame_num = sh.frame_num;
            sl.field_coded = sh.field_pic != 0;
        }
        const int pic_poc = compute_poc(s, sps, sh);
        if (sh.field_pic) {
            sl.fpoc[sh.bottom_field ? 1 : 0] = pic_poc;
            sl.fpic[sh.bottom_field ? 1 : 0] = s.cur_pic;
            if (!second) sl.poc = pic_poc;
        } else {
            sl.poc = pic_poc, sl.fpoc[0] = s.poc_top, sl.fpoc[1] = s.poc_bot;
            sl.fields = 3; // (a frame picture delivers both fields; it is not in its own reference lists)
            sl.pic = s.cur_pic;
        }
        PicDesc &pd = g.h_pics[s.cur_pic];
        memset(&pd, 0, sizeof(pd));
        pd.stream = si, pd.slot = slot, pd.wmb = wmb, pd.hmb = hmb_pic;
        // where the picture lives in its frame slot: a field picture in the rows of its parity (PicDesc)
        pd.field = static_cast<uint8_t>(s.cur_field);
        pd.pitch = static_cast<uint32_t>(wmb * 16 * (sh.field_pic ? 2 : 1)), pd.plane = static_cast<uint32_t>(wmb * 16) * static_cast<uint32_t>(hmb * 16);
        pd.inv_wmb = static_cast<uint32_t>((1ull << 32) / static_cast<uint32_t>(wmb)) + 1u;
        pd.pool_base = d->h_pools[si].base, pd.slot_bytes = d->slot_bytes, pd.n_slots = static_cast<uint32_t>(d->n_slots);
        g.pic_level.resize(g.n_pics, 0), g.pic_save_col.resize(g.n_pics, 0), g.pic_wave.resize(g.n_pics, 0);
        g.pic_level[s.cur_pic] = 0, g.pic_save_col[s.cur_pic] = 0, g.pic_wave[s.cur_pic] = 0;
        pd.mb_base = g.mb_used;
        g.mb_used += static_cast<uint64_t>(wmb) * hmb_pic;
        pd.first_slice = g.n_slices;
        pd.cabac = pps.entropy_coding_mode, pd.t8x8_mode = pps.transform_8x8_mode, pd.cip = pps.constrained_intra_pred;
        pd.mono = sps.chroma_format == 0;
        pd.weighted_pred = pps.weighted_pred;
        pd.cqp_off[0] = static_cast<int8_t>(pps.chroma_qp_index_offset), pd.cqp_off[1] = static_cast<int8_t>(pps.second_chroma_qp_index_offset);
        pd.is_intra_only = 1;
        int ss = scaling_set_for(d, pps);
        if (ss < 0) {
            set_error("more than %d distinct scaling matrices in flight", MI_MAX_SCALING_SETS);
            return H264MI_ECAPACITY;
        }
        pd.scaling_set = static_cast<uint8_t>(ss);
        pd.order = s.n_pics_in_batch++;
        if (pps.num_slice_groups_minus1 > 0) { // FMO: this picture's macroblock-to-slice-group map travels with the bitstream (8.2.2; h264/slice.go:134-158)
            const size_t n_mbs = static_cast<size_t>(wmb) * hmb_pic, moff = (g.map_cursor + 15) & ~static_cast<size_t>(15);
            if (moff + n_mbs + 4096 > d->bits_cap) {
                set_error("bitstream staging buffer too small for the slice group maps (%zu bytes)", d->bits_cap);
                return H264MI_ECAPACITY;
            }
            r = mb_to_slice_group_map(&sps, &pps, s.sg_ids[pps_id].data(), s.sg_ids[pps_id].size(), sh.slice_group_change_cycle, sh.field_pic ? 1 : 0, g.h_bits + moff, n_mbs, nullptr);
            if (r != H264MI_OK) return r;
            pd.fmo = 1, pd.sgmap_off = static_cast<uint32_t>(moff);
            g.map_cursor = moff + n_mbs;
            g.bits_end = std::max(g.bits_end, g.map_cursor);
        }
        if (!second) { // the frame this picture belongs to, as it will be shown: pushed now (frame picture), or when its fields are through (finish_picture / flush_pending_field)
            OutFrame of{slot, wmb, hmb, 2 * sps.frame_crop_left_offset, 2 * (2 - sps.frame_mbs_only) * sps.frame_crop_top_offset, sps.width, sps.height, sl.poc, sh.frame_num,
                        sh.nal_ref_idc, sh.nal_unit_type == 5, s.cur_pic, sh.nal_unit_type == 5};
            if (sh.nal_ref_idc && sh.adaptive_ref_pic_marking_mode_flag)
                for (int k = 0; k < sh.n_memory_management_control_operations; k++)
                    if (sh.memory_management_control_operation[k] == 5) of.new_sequence = 1;
            if (sh.field_pic)
                s.pend_out = of;
            else
                g.out[si].push_back(of);
        }
        g.wmb_max = std::max(g.wmb_max, wmb);
        g.hmb_max = std::max(g.hmb_max, hmb_pic);
        g.mbs_max = std::max(g.mbs_max, wmb * hmb_pic);
        g.info.n_macroblocks += static_cast<int64_t>(wmb) * hmb_pic;
        if (si == 0) g.info.width = sps.width, g.info.height = sps.height, g.info.coded_width = wmb * 16, g.info.coded_height = hmb * 16;
    }
    if (s.cur_slices >= d->cfg.max_slices_per_frame) {
        set_error("stream %d: more than %d slices in a frame", si, d->cfg.max_slices_per_frame);
        return H264MI_ECAPACITY;
    }
    if (sh.first_mb_in_slice >= wmb * hmb_pic) return H264MI_EBITSTREAM;
    s.cur_first_mbs.push_back(sh.first_mb_in_slice);
    PicDesc &pd = g.h_pics[s.cur_pic];
    SliceDesc &sd = g.h_slices[g.n_slices];
    memset(&sd, 0, sizeof(sd));
    sd.rbsp_off = static_cast<uint32_t>(off), sd.rbsp_size = static_cast<uint32_t>(rlen);
    sd.data_bit_off = static_cast<uint32_t>(sh.slice_data_bit_offset);
    {
        size_t n = rlen;
        while (n > 0 && rbsp[n - 1] == 0) n--;
        sd.stop_bit = n ? static_cast<uint32_t>((n - 1) * 8 + 7 - __builtin_ctz(rbsp[n - 1])) : 0;
    }
    sd.pic_idx = s.cur_pic, sd.first_mb = sh.first_mb_in_slice;
    sd.slice_type = static_cast<uint8_t>(st);
    sd.cabac_init_idc = static_cast<uint8_t>(sh.cabac_init), sd.slice_qp = static_cast<uint8_t>(sh.slice_qp_y);
    sd.num_ref_idx_active = static_cast<uint8_t>(st != 2 ? sh.num_ref_idx_l0_active_minus1 + 1 : 0);
    sd.alpha_off = static_cast<int8_t>(2 * sh.slice_alpha_c0_offset_div2), sd.beta_off = static_cast<int8_t>(2 * sh.slice_beta_offset_div2);
    sd.dbf_idc = static_cast<uint8_t>(sh.disable_deblocking_filter);
    sd.slice_in_pic = static_cast<uint16_t>(s.cur_slices);
    for (int i = 0; i < MI_MAX_REFS; i++) sd.ref_slot[i] = -1;
    int level = 0;
    if (st != 2) {
        pd.is_intra_only = 0;
        const bool bslice = st == 1;
        const bool explicit_wp = bslice ? pps.weighted_bipred == 1 : pps.weighted_pred != 0;
        BSliceExt bx;
        memset(&bx, 0, sizeof(bx));
        r = build_ref_lists(s, sps, sh, bslice, sd.ref_slot, bx.ref_slot1);
        if (r != H264MI_OK) return r;
        { // the pictures of this batch the slice predicts from have to be reconstructed (deblocked) first: Stage::pic_wave
            int wave = g.pic_wave[s.cur_pic];
            auto after = [&](int entry) {
                if (entry < 0) return;
                const Slot &rs = s.slots[sh.field_pic ? MI_REF_SLOT(entry) : entry];
                for (int pic : {rs.pic, rs.fpic[0], rs.fpic[1]}) // (a frame, or either field: whichever of them this batch decodes)
                    if (pic >= 0 && pic != s.cur_pic && pic < static_cast<int>(g.pic_wave.size())) wave = std::max(wave, g.pic_wave[pic] + 1);
            };
            for (int i = 0; i <= sh.num_ref_idx_l0_active_minus1 && i < MI_MAX_REFS; i++) after(sd.ref_slot[i]);
            if (bslice)
                for (int i = 0; i <= sh.num_ref_idx_l1_active_minus1 && i < MI_MAX_REFS; i++) after(bx.ref_slot1[i]);
            g.pic_wave[s.cur_pic] = wave;
        }
        sd.wp_flag = static_cast<uint8_t>(explicit_wp);
        sd.luma_log2_denom = static_cast<uint8_t>(sh.luma_log2_weight_denom), sd.chroma_log2_denom = static_cast<uint8_t>(sh.chroma_log2_weight_denom);
        for (int i = 0; i < MI_MAX_REFS; i++) {
            sd.wp_lw[i] = static_cast<int16_t>(explicit_wp ? sh.luma_weight_l0[i] : 1), sd.wp_lo[i] = static_cast<int16_t>(sh.luma_offset_l0[i]);
            for (int j = 0; j < 2; j++)
                sd.wp_cw[i][j] = static_cast<int16_t>(explicit_wp ? sh.chroma_weight_l0[i][j] : 1), sd.wp_co[i][j] = static_cast<int16_t>(sh.chroma_offset_l0[i][j]);
        }
        if (bslice) {
            r = ensure_b_buffers(d);
            if (r != H264MI_OK) return r;
            pd.has_b = 1;
            // PicOrderCnt of the current picture and of a list entry: a field's own count in a field picture (entries name fields), the frame's otherwise
            const bool fieldpic = sh.field_pic != 0;
            const int cur_poc = fieldpic ? s.slots[s.cur_slot].fpoc[sh.bottom_field ? 1 : 0] : s.slots[s.cur_slot].poc;
            auto entry_slot = [&](int e) -> const Slot & { return s.slots[fieldpic ? MI_REF_SLOT(e) : e]; };
            auto entry_poc = [&](int e) { return fieldpic ? entry_slot(e).fpoc[(e & MI_REF_PARITY) ? 1 : 0] : entry_slot(e).poc; };
            const int n0 = sh.num_ref_idx_l0_active_minus1 + 1, n1 = sh.num_ref_idx_l1_active_minus1 + 1;
            bx.num_ref_idx_l1_active = static_cast<uint8_t>(n1);
            bx.direct_spatial = static_cast<uint8_t>(sh.direct_spatial_mv_pred), bx.direct_8x8_inference = static_cast<uint8_t>(sps.direct_8x8_inference);
            bx.wp_mode = static_cast<uint8_t>(pps.weighted_bipred);
            for (int i = 0; i < MI_MAX_REFS; i++) {
                bx.wp_lw1[i] = static_cast<int16_t>(explicit_wp ? sh.luma_weight_l1[i] : 1), bx.wp_lo1[i] = static_cast<int16_t>(sh.luma_offset_l1[i]);
                for (int j = 0; j < 2; j++)
                    bx.wp_cw1[i][j] = static_cast<int16_t>(explicit_wp ? sh.chroma_weight_l1[i][j] : 1), bx.wp_co1[i][j] = static_cast<int16_t>(sh.chroma_offset_l1[i][j]);
            }
            // POC distances: DistScaleFactor of temporal direct prediction (8.4.1.2.3) per refIdxL0 against RefPicList1[0], and the
            // implicit bi-prediction weights (8.4.2.3.1) per (refIdxL0, refIdxL1)
            auto dist_scale = [&](int slot0, int slot1, bool *copy) {
                const Slot &p0 = entry_slot(slot0);
                const int poc0 = entry_poc(slot0), poc1 = entry_poc(slot1);
                const int tb = std::min(std::max(cur_poc - poc0, -128), 127), td = std::min(std::max(poc1 - poc0, -128), 127);
                *copy = td == 0 || p0.ref == 2;
                if (td == 0) return 256;
                const int tx = (16384 + std::abs(td / 2)) / td;
                return std::min(std::max((tb * tx + 32) >> 6, -1024), 1023);
            };
            const int col_slot = bx.ref_slot1[0];
            for (int i = 0; i < MI_MAX_REFS; i++) {
                bool copy = true;
                int dsf = 256;
                if (i < n0 && sd.ref_slot[i] >= 0 && col_slot >= 0) dsf = dist_scale(sd.ref_slot[i], col_slot, &copy);
                bx.dist_scale[i] = static_cast<int16_t>(copy ? 256 : dsf);
                for (int j = 0; j < MI_MAX_REFS; j++) {
                    int w1 = 32;
                    if (i < n0 && j < n1 && sd.ref_slot[i] >= 0 && bx.ref_slot1[j] >= 0) {
                        bool cp;
                        const int f = dist_scale(sd.ref_slot[i], bx.ref_slot1[j], &cp) >> 2;
                        const int td = entry_poc(bx.ref_slot1[j]) - entry_poc(sd.ref_slot[i]);
                        if (td != 0 && entry_slot(sd.ref_slot[i]).ref != 2 && entry_slot(bx.ref_slot1[j]).ref != 2 && f >= -64 && f <= 128) w1 = f;
                    }
                    bx.implicit_w1[i][j] = static_cast<int16_t>(w1);
                }
            }
            // the co-located picture: its motion record array, and when its slices are entropy-decoded relative to this one
            level = 1;
            if (col_slot >= 0) {
                const Slot &cs = entry_slot(col_slot);
                const int cslot = fieldpic ? MI_REF_SLOT(col_slot) : col_slot, cpar = fieldpic && (col_slot & MI_REF_PARITY) ? 1 : 0;
                // the co-located picture must have the structure of the current one (8.4.1.2.1's frame-from-field-pair and field-from-frame
                // cases, with their vertical vector scaling, are not implemented)
                if (cs.field_coded != fieldpic && !cs.nonexisting) {
                    set_error("stream %d: a B %s whose RefPicList1[0] was coded as %s is out of scope (direct prediction across picture structures)", si,
                              fieldpic ? "field" : "frame picture", cs.field_coded ? "field pictures" : "a frame picture");
                    return H264MI_EUNSUPPORTED;
                }
                // a field's motion: the second half of the frame slot's array for the bottom field (a field has half the frame's macroblocks)
                bx.col = reinterpret_cast<uint64_t>(d->d_colrec + (static_cast<size_t>(si) * d->n_slots + cslot) * d->colrec_per_slot + (cpar ? d->colrec_per_slot / 2 : 0));
                bx.col_short = cs.ref == 1;
                const int cpic = fieldpic ? cs.fpic[cpar] : cs.pic;
                if (cpic >= 0) {
                    level = g.pic_level[cpic] + 1;
                    g.pic_save_col[cpic] = 1;
                } else if (!cs.col_valid[cpar] && !cs.nonexisting) {
                    // decoded by an earlier batch, before this decoder kept motion (the arrays exist from the first B slice on, ensure_b_buffers):
                    // direct prediction from it would silently see an intra picture
                    set_error("stream %d: the co-located picture of a B slice was decoded before the decoder's first B slice; its motion was not kept", si);
                    return H264MI_EUNSUPPORTED;
                }
            }
            sd.bext = static_cast<uint32_t>(g.n_bext);
            g.h_bext[g.n_bext++] = bx;
        }
    }
    g.pic_level[s.cur_pic] = std::max(g.pic_level[s.cur_pic], level);
    g.slice_level.resize(g.n_slices + 1);
    g.slice_level[g.n_slices] = level;
    g.bits_used = std::max(g.bits_used, off + rlen);
    g.bits_end = std::max(g.bits_end, g.bits_used);
    g.n_slices++;
    s.cur_slices++;
    return H264MI_OK;
}

// What the entropy kernels reported for the slices of an executed batch (its status copy has arrived: the caller waited for
// ev_done or synchronised the streams).  A failed slice marks its stream: the pictures from there on are damaged, its references
// are dropped and nothing of it is decoded before its next IDR picture.  Called from h264mi_batch_sync for every batch that
// has been executed since the last look, and from h264mi_batch_prepare when it takes a staging set back -- so in the pipelined
// pattern (execute(k); prepare(k + 1); execute(k + 1); ...) a failure of batch k is acted upon before batch k + 2 is parsed.
static int harvest_status(h264mi_decoder *d, Stage &g) {
    if (!g.executed || g.harvested) return H264MI_OK;
    g.harvested = true;
    int result = H264MI_OK;
    for (int i = 0; i < g.n_slices; i++)
        if (g.h_status[8 * i]) {
            const SliceDesc &sd = g.h_slices[i];
            StreamState &s = d->st[g.h_pics[sd.pic_idx].stream];
            // the stream was reset (h264mi_decoder_reset / h264mi_stream_reset: a new connection took the slot, or the caller started over) after this
            // batch was prepared: what failed in it is not a property of what the slot decodes now
            if (g.h_pics[sd.pic_idx].stream < g.epochs.size() && g.epochs[g.h_pics[sd.pic_idx].stream] != s.epoch) continue;
            const bool field_cabac = g.h_pics[sd.pic_idx].field != 0 && g.h_pics[sd.pic_idx].cabac != 0;
            if (field_cabac) d->unpinned_failures++; // what a wrong value in the unpinned context tables looks like: the slice does not end on end_of_slice_flag where it should
            if (result == H264MI_OK)
                set_error("entropy kernel: slice %d (picture %u, stream %u) failed with code %u after %u macroblocks%s", i, sd.pic_idx, g.h_pics[sd.pic_idx].stream,
                          g.h_status[8 * i], g.h_status[8 * i + 1],
                          field_cabac ? " -- a CABAC field picture: the context values of field-coded blocks are unpinned (h264mi_config.allow_unpinned_field_cabac)" : "");
            if (s.status == H264MI_OK || s.status_batch != &g) { // first failure of the stream in this batch
                s.status = H264MI_EDECODE, s.status_batch = &g;
                for (auto &sl : s.slots) sl.ref = 0;
                s.need_idr = true;
            }
            result = H264MI_EDECODE;
        }
    return result;
}

extern "C" int32_t h264mi_batch_prepare(h264mi_decoder *d, int32_t n_streams, const uint8_t *const *bufs, const size_t *lens, h264mi_batch_info *info) {
    if (!d || n_streams < 0 || n_streams > static_cast<int>(d->st.size()) || (n_streams && (!bufs || !lens))) return H264MI_EINVAL;
    GUARD(d);
    auto t0 = std::chrono::steady_clock::now();
    // The next staging set; the batch that used it last (MI_STAGES batches ago) must have finished executing.  The batch
    // prepared before this one may still be executing: nothing it uses is touched here.
    const int prev_stage = d->prep;
    d->prep = (d->prep + 1) % MI_STAGES;
    Stage &g = d->stage[d->prep];
    if (g.executed) {
        HIP_TRY(hipEventSynchronize(g.ev_done));
        (void)harvest_status(d, g); // a batch nobody synchronised on: its failures still mark their streams (need_idr) before this parse
    }
    g.prepared = false, g.executed = false;
    g.n_slices = g.n_pics = 0;
    g.bits_used = 0, g.mb_used = 0, g.wmb_max = 0, g.hmb_max = 0, g.mbs_max = 0;
    g.map_cursor = g.bits_end = 0;
    g.fmo_pics.clear();
    g.grey.clear();
    g.epochs.resize(d->st.size());
    for (size_t si = 0; si < d->st.size(); si++) g.epochs[si] = d->st[si].epoch;
    g.n_bext = 0;
    g.pic_level.clear(), g.slice_level.clear(), g.pic_save_col.clear(), g.pic_wave.clear();
    memset(&g.info, 0, sizeof(g.info));
    for (size_t si = 0; si < d->st.size(); si++) {
        StreamState &s = d->st[si];
        for (auto &sl : s.slots) sl.held = sl.ref != 0, sl.pic = sl.fpic[0] = sl.fpic[1] = -1; // reference pictures at batch start stay put for the whole batch
        if (s.pend_slot >= 0) s.slots[s.pend_slot].held = true, s.pend_out.pic = -1; // a first field waiting for its second one (decoded by an earlier batch now)
        // the frames of the batch prepared before this one stay readable (and, if it is still executing, writable)
        if (MI_STAGES > 1)
            for (const OutFrame &o : d->stage[prev_stage].out[si]) s.slots[o.slot].held = true;
        g.out[si].clear();
        s.n_pics_in_batch = 0;
        s.cur_slot = s.cur_pic = -1;
        s.status = H264MI_OK;
    }
    // ---- pass 1 (parallel over streams): Annex-B scan ----
    const int n_threads = std::max(1, std::min<int>({static_cast<int>(std::thread::hardware_concurrency()), 16, n_streams}));
    std::vector<std::vector<h264mi_nal>> all_nals(n_streams);
    std::vector<int> n_nals(n_streams, 0);
    auto parallel_for = [&](int count, const std::function<void(int)> &fn) {
        if (n_threads <= 1 || count <= 1) {
            for (int i = 0; i < count; i++) fn(i);
            return;
        }
        std::atomic<int> next(0);
        std::vector<std::thread> pool;
        for (int t = 0; t < std::min(n_threads, count); t++)
            pool.emplace_back([&] {
                for (int i = next.fetch_add(1); i < count; i = next.fetch_add(1)) fn(i);
            });
        for (auto &t : pool) t.join();
    };
    parallel_for(n_streams, [&](int si) {
        if (!bufs[si] || !lens[si]) return;
        std::vector<h264mi_nal> &v = all_nals[si];
        v.resize(1024);
        int n = 0;
        while (annexb_scan(bufs[si], lens[si], v.data(), static_cast<int>(v.size()), &n) == H264MI_ECAPACITY) v.resize(v.size() * 4);
        n_nals[si] = n;
    });
    // ---- pass 2 (serial, trivial): staging offsets of the slice NALs; the escaped length bounds the RBSP length ----
    struct Staged {
        int si, nal;
        size_t off, rlen;
    };
    std::vector<Staged> staged;
    std::vector<std::vector<int>> staged_of(n_streams); // [stream][nal] -> index into staged or -1
    {
        size_t cursor = 0;
        for (int si = 0; si < n_streams; si++) {
            staged_of[si].assign(n_nals[si], -1);
            for (int i = 0; i < n_nals[si]; i++) {
                const h264mi_nal &nal = all_nals[si][i];
                if (nal.type != 1 && nal.type != 5) continue;
                const size_t off = (cursor + 15) & ~static_cast<size_t>(15);
                if (off + nal.num_bytes + 4096 > d->bits_cap) {
                    set_error("bitstream staging buffer too small (%zu bytes)", d->bits_cap);
                    return H264MI_ECAPACITY;
                }
                staged_of[si][i] = static_cast<int>(staged.size());
                staged.push_back({si, i, off, 0});
                cursor = off + nal.num_bytes;
            }
        }
        g.map_cursor = cursor; // slice group maps (FMO pictures) go behind the last slice
    }
    // ---- pass 3 (parallel over slices): remove emulation prevention straight into the pinned staging buffer ----
    parallel_for(static_cast<int>(staged.size()), [&](int k) {
        Staged &sg = staged[k];
        const h264mi_nal &nal = all_nals[sg.si][sg.nal];
        sg.rlen = unescape(bufs[sg.si] + nal.offset + 1, nal.num_bytes - 1, g.h_bits + sg.off);
    });
    // ---- pass 4 (serial): parameter sets, slice headers, DPB / POC / reference lists, descriptors ----
    int first_error = H264MI_OK;
    for (int si = 0; si < n_streams; si++) {
        if (!bufs[si] || !lens[si]) continue;
        StreamState &s = d->st[si];
        const std::vector<h264mi_nal> &nals = all_nals[si];
        const int n = n_nals[si];
        std::vector<uint8_t> tmp;
        // what this stream adds to the batch sits at the tail of every table: a failing stream is taken out again
        const int pics0 = g.n_pics, slices0 = g.n_slices, bext0 = g.n_bext;
        const uint64_t mb0 = g.mb_used;
        const int64_t info_mb0 = g.info.n_macroblocks;
        int r = H264MI_OK;
        for (int i = 0; i < n && r == H264MI_OK; i++) {
            const h264mi_nal &nal = nals[i];
            const uint8_t *p = bufs[si] + nal.offset;
            switch (nal.type) {
            case 7: {
                tmp.resize(nal.num_bytes);
                size_t rl = unescape(p + 1, nal.num_bytes - 1, tmp.data());
                h264mi_sps sps;
                r = parse_sps(tmp.data(), rl, &sps);
                if (r == H264MI_OK) {
                    if (s.cur_slot >= 0) finish_picture(d, si);
                    s.sps[sps.id] = sps, s.sps_ok[sps.id] = true;
                }
                break;
            }
            case 8: {
                tmp.resize(nal.num_bytes);
                size_t rl = unescape(p + 1, nal.num_bytes - 1, tmp.data());
                BitReader br(tmp.data(), rl);
                br.ue();
                uint32_t sid = br.ue();
                if (sid > 31 || !s.sps_ok[sid]) {
                    set_error("stream %d: PPS refers to missing SPS %u", si, sid);
                    r = H264MI_EBITSTREAM;
                    break;
                }
                h264mi_pps pps;
                std::vector<uint8_t> ids(static_cast<size_t>(s.sps[sid].pic_width_in_mbs_minus1 + 1) * (s.sps[sid].pic_height_in_map_units_minus1 + 1));
                size_t n_ids = 0;
                r = parse_pps_ids(&s.sps[sid], tmp.data(), rl, &pps, ids.data(), ids.size(), &n_ids);
                if (r == H264MI_OK) {
                    if (s.cur_slot >= 0) finish_picture(d, si);
                    s.pps[pps.id] = pps, s.pps_ok[pps.id] = true;
                    ids.resize(n_ids);
                    s.sg_ids[pps.id] = std::move(ids);
                }
                break;
            }
            case 1:
            case 5: {
                const Staged &sg = staged[staged_of[si][i]];
                r = add_slice(d, si, sg.off, sg.rlen, nal.ref_idc, nal.type);
                break;
            }
            case 9:
            case 10:
            case 11:
                if (s.cur_slot >= 0) finish_picture(d, si);
                if (nal.type != 9) flush_pending_field(d, si); // end of sequence / end of stream: no second field will follow a lone first one
                break;
            case 2:
            case 3:
            case 4: // slice data partitions A / B / C (Extended profile; h264/nalUnit.go:14-16 names them): ignoring them would drop pictures without a word
                set_error("stream %d: slice data partitioning (nal_unit_type %d, Extended profile) is out of scope", si, nal.type);
                r = H264MI_EUNSUPPORTED;
                break;
            default: break; // SEI, filler, the extension units of SVC / MVC / 3D-AVC streams (14, 15, 20, 21: the base layer is decoded) ... (h264/server.go:147-164 ignores them too)
            }
        }
        if (r != H264MI_OK) {
            // The stream leaves the batch: its pictures, slices and records are dropped, its references are forgotten
            // (nothing is decodable before its next IDR picture); the other streams are not affected.
            g.n_pics = pics0, g.n_slices = slices0, g.mb_used = mb0, g.info.n_macroblocks = info_mb0;
            g.n_bext = bext0;
            g.pic_level.resize(pics0), g.pic_save_col.resize(pics0), g.pic_wave.resize(pics0), g.slice_level.resize(slices0);
            while (!g.fmo_pics.empty() && static_cast<int>(g.fmo_pics.back()) >= pics0) g.fmo_pics.pop_back();
            g.grey.erase(std::remove_if(g.grey.begin(), g.grey.end(), [&](const Stage::GreyFill &f) { return static_cast<int>(f.stream) == si; }), g.grey.end());
            reset_stream(s, true);
            g.out[si].clear();
            s.need_idr = true;
            s.status = r;
            if (first_error == H264MI_OK) first_error = r;
            if (!d->isolate) return r;
            continue;
        }
        if (s.cur_slot >= 0) finish_picture(d, si);
    }
    // A reference picture that outlives the batch may become the co-located picture of a B picture of a later batch: its
    // motion is kept too (8.4.1.2.1).
    g.pic_level.resize(g.n_pics, 0), g.pic_save_col.resize(g.n_pics, 0), g.pic_wave.resize(g.n_pics, 0), g.slice_level.resize(g.n_slices, 0);
    for (int si = 0; si < n_streams; si++)
        for (const Slot &sl : d->st[si].slots)
            if (sl.ref)
                for (int pic : {sl.pic, sl.fpic[0], sl.fpic[1]})
                    if (pic >= 0 && pic < g.n_pics) g.pic_save_col[pic] = 1;
    // Slice order = launch order: by level (see Stage), and inside a level longest-processing-time-first -- the entropy
    // kernels run one slice per workgroup and workgroups are dispatched in index order, so the biggest slices (I pictures)
    // must start first.
    int n_levels = 1;
    {
        std::vector<int> order(g.n_slices);
        for (int i = 0; i < g.n_slices; i++) order[i] = i, n_levels = std::max(n_levels, g.slice_level[i] + 1);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
            if (g.slice_level[x] != g.slice_level[y]) return g.slice_level[x] < g.slice_level[y];
            return g.h_slices[x].rbsp_size > g.h_slices[y].rbsp_size;
        });
        std::vector<SliceDesc> tmp(g.h_slices, g.h_slices + g.n_slices);
        std::vector<int> lv(g.slice_level);
        for (int i = 0; i < g.n_slices; i++) g.h_slices[i] = tmp[order[i]], g.slice_level[i] = lv[order[i]];
        g.level_first.assign(n_levels + 1, g.n_slices);
        for (int i = g.n_slices - 1; i >= 0; i--) g.level_first[g.slice_level[i]] = i;
        for (int l = n_levels - 1; l >= 0; l--) g.level_first[l] = std::min(g.level_first[l], g.level_first[l + 1]);
    }
    // picture "waves": pictures that do not predict from one another are reconstructed side by side -- wave 0 holds the pictures that read no
    // picture of this batch (intra pictures wherever they stand in their stream; pictures whose references an earlier batch decoded), wave n + 1
    // the pictures whose latest reference is of wave n (Stage::pic_wave; with I P P P ... that is the k-th picture of every stream, with
    // I B B P ... the B pictures share the wave of the P picture that follows them in decoding order, an all-intra stream is one wave).  Per wave:
    // all pictures (K3), the inter pictures without / the pictures with B slices (K4 / K4 two-list), the pictures without B slices (K5; with: K5 two-list)
    size_t nw = 0;
    for (int i = 0; i < g.n_pics; i++) nw = std::max<size_t>(nw, static_cast<size_t>(g.pic_wave[i]) + 1);
    g.waves.assign(nw, {});
    g.waves_inter.assign(nw, {});
    for (int i = 0; i < g.n_pics; i++) {
        g.waves[g.pic_wave[i]].push_back(i);
        if (!g.h_pics[i].is_intra_only) g.waves_inter[g.pic_wave[i]].push_back(i);
    }
    g.wave_off.clear();
    g.wave_inter_off.clear();
    g.wave_b_off.assign(nw, 0), g.wave_b_n.assign(nw, 0), g.wave_p_off.assign(nw, 0), g.wave_p_n.assign(nw, 0), g.wave_nb_off.assign(nw, 0), g.wave_nb_n.assign(nw, 0);
    uint32_t pos = 0;
    for (size_t w = 0; w < nw; w++) {
        g.wave_off.push_back(pos);
        for (uint32_t p : g.waves[w]) g.h_lists[pos++] = p;
        g.wave_inter_off.push_back(pos);
        g.wave_p_off[w] = pos;
        for (uint32_t p : g.waves_inter[w])
            if (!g.h_pics[p].has_b) g.h_lists[pos++] = p;
        g.wave_p_n[w] = pos - g.wave_p_off[w];
        g.wave_b_off[w] = pos;
        for (uint32_t p : g.waves[w])
            if (g.h_pics[p].has_b) g.h_lists[pos++] = p;
        g.wave_b_n[w] = pos - g.wave_b_off[w];
        g.wave_nb_off[w] = pos;
        for (uint32_t p : g.waves[w])
            if (!g.h_pics[p].has_b) g.h_lists[pos++] = p;
        g.wave_nb_n[w] = pos - g.wave_nb_off[w];
    }
    g.colsave_n.assign(n_levels, 0), g.prep_off.assign(n_levels, 0), g.prep_n.assign(n_levels, 0);
    for (int l = 0; l < n_levels; l++) {
        g.prep_off[l] = pos;
        for (int i = 0; i < g.n_pics; i++)
            if (g.pic_level[i] == l) {
                g.h_lists[pos++] = i;
                PicDesc &pd = g.h_pics[i];
                pd.save_col = g.pic_save_col[i] && d->d_colrec != nullptr; // (no B slice seen yet: nothing to keep, see ensure_b_buffers)
                const int par = pd.field == 2 ? 1 : 0; // (a bottom field's motion: the second half of its frame slot's array)
                pd.col_out = d->d_colrec ? reinterpret_cast<uint64_t>(d->d_colrec + (static_cast<size_t>(pd.stream) * d->n_slots + pd.slot) * d->colrec_per_slot +
                                                                      (par ? d->colrec_per_slot / 2 : 0))
                                         : 0;
                if (pd.save_col) d->st[pd.stream].slots[pd.slot].col_valid[par] = true; // (the slot is this picture's until the next prepare at least: held)
                g.colsave_n[l] += pd.save_col;
            }
        g.prep_n[l] = pos - g.prep_off[l];
    }
    // Uploads go to the entropy stream the next execute will use: in order with that pass's entropy kernel, and not behind
    // the batch that is still executing (the caller's stream waits for its reconstruction).  No stream of their own: the
    // runtime multiplexes streams onto 4 hardware queues by default, and a fifth stream ended up sharing a queue with one of
    // the kernel streams, which serialised entropy decoding and reconstruction of consecutive passes.
    hipStream_t up = d->ent_stream[d->pass & 1];
    if (g.n_slices) {
        const size_t end = std::max(g.bits_used, g.bits_end); // (bits_end: behind the slice group maps of FMO pictures, if any)
        size_t nbytes = std::min(d->bits_cap, ((end + 15) & ~static_cast<size_t>(15)) + 4096);
        memset(g.h_bits + end, 0, nbytes - end);
        HIP_TRY(hipMemcpyAsync(g.d_bits, g.h_bits, nbytes, hipMemcpyHostToDevice, up));
        HIP_TRY(hipMemcpyAsync(g.d_slices, g.h_slices, sizeof(SliceDesc) * g.n_slices, hipMemcpyHostToDevice, up));
        HIP_TRY(hipMemcpyAsync(g.d_pics, g.h_pics, sizeof(PicDesc) * g.n_pics, hipMemcpyHostToDevice, up));
        HIP_TRY(hipMemcpyAsync(g.d_lists, g.h_lists, sizeof(uint32_t) * pos, hipMemcpyHostToDevice, up));
        if (g.n_bext) HIP_TRY(hipMemcpyAsync(g.d_bext, g.h_bext, sizeof(BSliceExt) * g.n_bext, hipMemcpyHostToDevice, up));
    }
    if (d->tables_dirty) {
        // a new PPS added a LevelScale set: the table only grows, so the batch still executing keeps seeing its own sets.
        // (h_tables is pinned and rewritten only by the next prepare, which cannot start its copy before this one is done:
        // same stream.)
        HIP_TRY(hipMemcpyAsync(d->d_tables, d->h_tables, sizeof(DevTables), hipMemcpyHostToDevice, up));
        d->tables_dirty = false;
    }
    HIP_TRY(hipEventRecord(g.ev_upload, up));
    g.info.n_frames = 0; // frames that go out with this batch (a frame coded as two field pictures counts once, where its second field is)
    for (const auto &o : g.out) g.info.n_frames += static_cast<int32_t>(o.size());
    g.info.n_slices = g.n_slices, g.info.bitstream_bytes = static_cast<int64_t>(g.bits_used);
    g.info.host_prepare_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (info) *info = g.info;
    g.prepared = true;
    return H264MI_OK;
}

static hipEvent_t next_event(h264mi_decoder *d, size_t &idx, int kind) {
    if (idx >= d->ev.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        d->ev.push_back(e);
        d->ev_kind.push_back(kind);
    }
    d->ev_kind[idx] = kind;
    return d->ev[idx++];
}

// One pass over a prepared batch.  exclusive: the pass runs alone (the caller has drained every stream), serialised on the decoder's stream, with
// record set 0 and ALL of the residual pool -- how a batch that ran out of residual blocks is repeated (retry_exhausted).
static int execute_stage(h264mi_decoder *d, int stage_idx, bool exclusive);

extern "C" int32_t h264mi_batch_execute(h264mi_decoder *d) {
    if (!d || !d->stage[d->prep].prepared) {
        set_error("h264mi_batch_execute: no prepared batch");
        return H264MI_EINVAL;
    }
    d->exec = d->prep; // sync and the frame accessors refer to this batch from now on
    d->stage[d->exec].retried = false;
    return execute_stage(d, d->exec, false);
}

static int execute_stage(h264mi_decoder *d, int stage_idx, bool exclusive) {
    if (exclusive) d->pass += (MI_SETS - d->pass % MI_SETS) % MI_SETS; // record set 0
    d->last_exec_stage = stage_idx, d->last_exec_set = static_cast<int>(d->pass % MI_SETS);
    Stage &g = d->stage[stage_idx];
    if (!g.n_slices && g.grey.empty()) return H264MI_OK;
    GUARD(d);
    // frames that went out with one field decoded (flush_pending_field): the rows of the field that never came are painted mid-grey
    auto grey_fills = [&](hipStream_t st) -> hipError_t { // the first error, if any: a frame that was not painted must not go out as if it were
        for (const Stage::GreyFill &f : g.grey) {
            uint8_t *base = d->d_frames + (static_cast<size_t>(f.stream) * d->n_slots + f.slot) * d->slot_bytes;
            const size_t W = f.w, H = f.h, plane = W * H;
            hipError_t e = hipMemset2DAsync(base + f.parity * W, 2 * W, 128, W, H / 2, st);
            if (e == hipSuccess) e = hipMemset2DAsync(base + plane + f.parity * (W / 2), W, 128, W / 2, H / 4, st);
            if (e == hipSuccess) e = hipMemset2DAsync(base + plane + plane / 4 + f.parity * (W / 2), W, 128, W / 2, H / 4, st);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    if (!g.n_slices) { // nothing to decode: a chunk that only ended a sequence and thereby sent a lone first field out
        HIP_TRY(grey_fills(d->stream));
        HIP_TRY(hipEventRecord(g.ev_done, d->stream));
        g.executed = true, g.harvested = true;
        return H264MI_OK;
    }
    size_t ei = 0;
    const bool prof = d->profiling || exclusive; // (an exclusive pass takes the serialised path of the profiling mode; its event marks are harmless)
    const uint32_t pool_blocks = static_cast<uint32_t>(exclusive ? std::min<uint64_t>(d->pool_blocks * MI_SETS, 0xFFFF0000ull) : d->pool_blocks);
    auto mark = [&](int kind) {
        if (prof) hipEventRecord(next_event(d, ei, kind), d->stream);
    };
    // Pass n uses buffer set n&1.  Entropy runs on its own stream so that the entropy kernels of
    // pass n+1 overlap the reconstruction kernels of pass n; set reuse is fenced by events.
    const int set = static_cast<int>(d->pass % MI_SETS);
    hipStream_t es = d->ent_stream[d->pass & 1];
    MbRec *mbrec = d->d_mbrec[set];
    int16_t *coef = d->d_coef[set];
    // Entropy decoding, level by level (Stage::level_first): k_entropy for the I / P slices, k_entropy_b for each level of B
    // slices, and after each level k_dbprep for the pictures complete with it (K5's parameters; the ColRec arrays later B slices ask for).
    // `done`: recorded after the last level's k_dbprep -- what the reconstruction kernels wait for.
    auto launch_entropy = [&](hipStream_t st, size_t lds_pad, bool fence_prev_pass, hipEvent_t done) {
        const int n_levels = static_cast<int>(g.level_first.size()) - 1;
        // pictures with slice groups: a slice's macroblocks are scattered, so its wavefront cannot blank "its" range for what it
        // does not decode (SliceDesc::fill_from) -- all records of such a picture start out as MBT_NONE instead
        for (uint32_t pi : g.fmo_pics) hipMemsetAsync(mbrec + g.h_pics[pi].mb_base, 0, sizeof(MbRec) * g.h_pics[pi].wmb * g.h_pics[pi].hmb, st);
        hipMemsetAsync(d->d_imask[set], 0, sizeof(unsigned long long) * (g.mb_used / 64 + 2), st); // k_dbprep ORs K3's work list into it
        for (int lv = 0; lv < n_levels; lv++) {
            const int first = g.level_first[lv], n = g.level_first[lv + 1] - first;
            // ColRec arrays cross passes: B slices read what the previous pass's k_dbprep wrote, and this pass's k_dbprep may
            // rewrite the record array of a frame slot (released meanwhile) that the previous pass's B slices still read.  So
            // everything after the I/P launch waits for the previous pass's entropy stream -- the I/P launch itself does not
            // (it neither reads nor writes ColRec arrays), it overlaps the previous pass's B launches; batches without B slices on
            // both sides never wait.
            auto fence = [&] {
                if (fence_prev_pass) hipStreamWaitEvent(st, d->ev_col[(d->pass - 1) % MI_SETS], 0);
                fence_prev_pass = false;
            };
            if (lv > 0) fence();
            if (n > 0) {
                if (lv == 0)
                    hipLaunchKernelGGL(g.fmo_pics.empty() ? k_entropy : k_entropy_f, dim3(n), dim3(64), lds_pad, st, g.d_slices, g.d_pics, g.d_bits, d->d_tables, mbrec, coef, d->d_pool_head + set,
                                       pool_blocks, g.d_status, d->d_toprows[set], g.wmb_max, static_cast<uint32_t>(first));
                else
                    hipLaunchKernelGGL(k_entropy_b, dim3(n), dim3(64), 0, st, g.d_slices, g.d_pics, g.d_bits, d->d_tables, mbrec, coef, d->d_pool_head + set,
                                       pool_blocks, g.d_status, d->d_toprows[set], g.wmb_max, static_cast<uint32_t>(first), g.d_bext, d->d_mv1[set]);
            }
            // the pictures complete with this level: K5's strengths and filter parameters, and the motion later B slices (the next
            // level's, or a later batch's) take their direct prediction from
            if (g.colsave_n[lv]) fence();
            if (g.prep_n[lv])
                hipLaunchKernelGGL(k_dbprep, dim3((g.mbs_max + MI_DBPREP_MBS - 1) / MI_DBPREP_MBS, (g.prep_n[lv] + 7u) & ~7u), dim3(256), 0, st, g.d_lists + g.prep_off[lv], g.d_pics,
                                   d->d_tables, mbrec, d->d_mv1[set], d->d_dbprm[set], 0, d->d_imask[set], static_cast<int>(g.prep_n[lv]));
            if (lv == n_levels - 1 && done) hipEventRecord(done, st);
        }
    };
    if (prof) { // profiling serialises the two stages on one stream so that HIP-event intervals are per kernel
        HIP_TRY(hipStreamWaitEvent(d->stream, g.ev_upload, 0));
        mark(-1);
        HIP_TRY(hipMemsetAsync(d->d_pool_head + set, 0, sizeof(uint32_t), d->stream));
        launch_entropy(d->stream, 0, false, nullptr);
        mark(0);
    } else {
        HIP_TRY(hipStreamWaitEvent(es, g.ev_upload, 0));
        if (d->pass >= MI_SETS) { // pass n-MI_SETS finished reading this set: its reconstruction kernels and its entropy stream
            HIP_TRY(hipStreamWaitEvent(es, d->ev_rec[set], 0));
            HIP_TRY(hipStreamWaitEvent(es, d->ev_col[set], 0));
        }
        HIP_TRY(hipMemsetAsync(d->d_pool_head + set, 0, sizeof(uint32_t), es));
        launch_entropy(es, d->ent_lds_pad, (g.n_bext || d->last_pass_had_b) && d->pass > 0, d->ev_ent[set]);
        d->last_pass_had_b = g.n_bext > 0;
        HIP_TRY(hipEventRecord(d->ev_col[set], es)); // entropy stream through with this pass
        HIP_TRY(hipStreamWaitEvent(d->rec_stream, d->ev_ent[set], 0));
    }
    hipStream_t rs = prof ? d->stream : d->rec_stream;
    if (d->last_pack && !prof) { // a pack launch may still be reading frames this pass rewrites (the same batch executed again)
        HIP_TRY(hipStreamWaitEvent(rs, d->last_pack, 0));
        d->last_pack = nullptr;
    }
    // tag of a banded launch's hand-off words: no earlier launch on the same rings has used it (the rings start out zero, and
    // are zeroed again should the counter ever wrap)
    auto next_epoch = [&]() {
        if (++d->x_epoch == 0) {
            hipMemsetAsync(d->d_xring, 0, static_cast<size_t>(d->x_cap) * (d->Wmax / 16) * 24 * sizeof(unsigned long long), rs);
            hipMemsetAsync(d->d_xdone, 0, static_cast<size_t>(d->x_cap3) * (d->Wmax / 16) * sizeof(uint32_t), rs);
            d->x_epoch = 1;
        }
        return d->x_epoch;
    };
    HIP_TRY(grey_fills(rs));
    for (size_t w = 0; w < g.waves.size(); w++) {
        const uint32_t n = static_cast<uint32_t>(g.waves[w].size()), ni = g.wave_p_n[w], nbp = g.wave_b_n[w];
        if (!n) continue;
        int grp_log2 = 0; // K4 workgroups per picture: the largest picture's count of four-macroblock groups, rounded up to a power of two
        while ((4 << grp_log2) < g.mbs_max) grp_log2++;
        if (ni) {
            const uint32_t nb = ni << grp_log2;
            hipLaunchKernelGGL(k_inter, dim3(((nb + MI_K4_WG - 1) / MI_K4_WG + 7) & ~7u), dim3(64 * MI_K4_WG), 0, rs, g.d_lists + g.wave_p_off[w], g.d_pics, g.d_slices, d->d_tables, mbrec, coef, grp_log2,
                               static_cast<int>(nb));
            mark(1);
        }
        if (nbp) {
            const uint32_t nb = nbp << grp_log2;
            hipLaunchKernelGGL(k_inter_b, dim3(((nb + MI_K4_WG - 1) / MI_K4_WG + 7) & ~7u), dim3(64 * MI_K4_WG), 0, rs, g.d_lists + g.wave_b_off[w], g.d_pics, g.d_slices, d->d_tables, mbrec, coef, grp_log2,
                               static_cast<int>(nb), g.d_bext, d->d_mv1[set]);
            mark(1);
        }
        // K3 / K5 keep a picture inside one workgroup when the launch has pictures enough to fill the chip; otherwise the
        // banded kernels spread each picture over up to x_max_wgs / pictures workgroups (mi_intra_bands / mi_deblock_bands)
        // (K3 of a launch WITHOUT intra-only pictures -- the few intra macroblocks of P / B pictures -- is bound by the row with the most
        // of them: there the banded kernel also puts several wavefronts on a row)
        int nb3 = 1, nw3 = MI_INTRA_WAVES, wpr3 = 1;
        bool has_ipic = false;
        for (uint32_t pi : g.waves[w]) has_ipic |= g.h_pics[pi].is_intra_only != 0;
        mi_intra_bands(static_cast<int>(n), g.hmb_max, d->x_max_wgs, !has_ipic, &nb3, &nw3, &wpr3);
        if (nb3 > 1) {
            hipLaunchKernelGGL(k_intra_x, dim3(n * nb3), dim3(nw3 * 64), 0, rs, g.d_lists + g.wave_off[w], g.d_pics, d->d_pools, d->d_tables, mbrec, coef, d->d_xdone,
                               next_epoch(), nb3, d->d_xctl + 32, d->x_tk3, g.wmb_max, d->d_xctl + 64, wpr3, d->d_imask[set]);
            d->x_tk3 += n * nb3;
        } else
            hipLaunchKernelGGL(k_intra, dim3(n), dim3(MI_INTRA_WAVES * 64), 0, rs, g.d_lists + g.wave_off[w], g.d_pics, d->d_pools, d->d_tables, mbrec, coef, d->d_imask[set]);
        mark(2);
        int dbw = 1, dbring = 16, dbring_last = 16, dbbufs = 1, nb5 = 1, roles = 1;
        mi_deblock_bands(static_cast<int>(n), g.wmb_max, g.hmb_max, d->x_max_wgs, &nb5, &dbw, &dbring, &roles);
        if (nb5 > 1) {
            hipLaunchKernelGGL(k_deblock_x, dim3(n * nb5), dim3(dbw * 64), mi_deblock_lds_bytes_banded(dbw, dbring), rs, g.d_lists + g.wave_off[w], g.d_pics,
                               d->d_dbprm[set], dbring, 0, 1, d->d_xring, next_epoch(), nb5, d->d_xctl, d->x_tk5, g.wmb_max, d->d_xctl + 64, roles);
            d->x_tk5 += n * nb5;
        } else {
            mi_deblock8_plan(g.wmb_max, g.hmb_max, &dbw, &dbring, &dbring_last, &dbbufs, d->k5_max_waves);
            hipLaunchKernelGGL(k_deblock, dim3(n), dim3(dbw * 64), mi_deblock8_lds_bytes(dbw, dbring, dbring_last, dbbufs), rs, g.d_lists + g.wave_off[w], g.d_pics,
                               d->d_dbprm[set], dbring, dbring_last, dbbufs, d->d_xctl + 64);
        }
        mark(3);
    }
    HIP_TRY(hipEventRecord(d->ev_rec[set], rs));
    if (!prof) HIP_TRY(hipStreamWaitEvent(d->stream, d->ev_rec[set], 0)); // the caller's stream sees the finished pass
    d->pass++;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(g.h_status, g.d_status, sizeof(uint32_t) * 8 * g.n_slices, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipMemcpyAsync(d->h_xstatus, d->d_xctl + 64, sizeof(uint32_t), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipEventRecord(g.ev_done, d->stream)); // d->stream has waited for the reconstruction kernels: the batch's buffers are idle after this
    g.executed = true, g.harvested = false;
    d->ev_used = d->profiling ? ei : 0;
    return H264MI_OK;
}

// A pass whose slices ran out of residual blocks (entropy status 40: the pool holds 8 blocks per macroblock by default, the worst case is 26) is
// repeated once, alone, with the pools of all record sets as one -- three times the room -- before anything is held against its streams.
// Only the batch executed LAST is repeated: a batch executed after the exhausted one has already written its pictures into frame slots the
// exhausted batch's first pictures predict from (slots that became free while it ran), so repeating the older batch would predict from the
// younger one's samples and report success; in that case the exhaustion is held against the streams as any other entropy failure
// (a caller that pipelines execute(n), prepare(n + 1), execute(n + 1), sync sizes its pool -- h264mi_config.coef_blocks_per_mb -- or
// synchronises per batch).  Called by h264mi_batch_sync with every stream drained.
static int retry_exhausted(h264mi_decoder *d) {
    Stage &g = d->stage[d->exec]; // the batch executed last
    if (!g.executed || g.harvested || g.retried) return H264MI_OK;
    bool exhausted = false;
    for (int i = 0; i < g.n_slices && !exhausted; i++) exhausted = g.h_status[8 * i] == 40;
    if (!exhausted) return H264MI_OK;
    g.retried = true;
    const int r = execute_stage(d, d->exec, true);
    if (r != H264MI_OK) return r;
    HIP_TRY(hipStreamSynchronize(d->stream));
    return H264MI_OK;
}

extern "C" int32_t h264mi_batch_sync(h264mi_decoder *d) {
    if (!d) return H264MI_EINVAL;
    GUARD(d);
    for (int i = 0; i < 2; i++) HIP_TRY(hipStreamSynchronize(d->ent_stream[i]));
    HIP_TRY(hipStreamSynchronize(d->rec_stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    if (d->profiling && d->ev_used >= 2) {
        size_t n = d->ev_used;
        double acc[4] = {0, 0, 0, 0};
        for (auto &v : d->launch_ms) v.clear();
        for (size_t i = 1; i < n && i < d->ev.size(); i++) {
            float ms = 0;
            hipEventElapsedTime(&ms, d->ev[i - 1], d->ev[i]);
            int k = d->ev_kind[i];
            if (k >= 0 && k < 4) acc[k] += ms, d->launch_ms[k].push_back(ms);
        }
        float tot = 0;
        hipEventElapsedTime(&tot, d->ev[0], d->ev[n - 1]);
        for (int k = 0; k < 4; k++) d->k_ms[k] = acc[k];
        d->k_ms[4] = tot;
    }
#if defined(H264MI_TEST_HOOKS)
    if (getenv("H264MI_SLICE_STATS")) { // diagnostics: per-slice entropy time (100 MHz ticks) and bin count (MI_ENT_STATS builds)
        Stage &g = d->stage[d->exec];
        for (int i = 0; i < g.n_slices && i < 64; i++)
            fprintf(stderr, "slice %d type %d bytes %u mbs %u us %.1f bins %u | Mclk fill %.1f syntax %.1f residual %.1f writeout %.1f\n", i, g.h_slices[i].slice_type,
                    g.h_slices[i].rbsp_size, g.h_status[8 * i + 1], g.h_status[8 * i + 2] * 0.01, g.h_status[8 * i + 3], g.h_status[8 * i + 4] * 16e-6,
                    g.h_status[8 * i + 5] * 16e-6, g.h_status[8 * i + 6] * 16e-6, g.h_status[8 * i + 7] * 16e-6);
        fprintf(stderr, "last slice type %d bytes %u mbs %u us %.1f bins %u\n", g.h_slices[g.n_slices - 1].slice_type, g.h_slices[g.n_slices - 1].rbsp_size,
                g.h_status[8 * (g.n_slices - 1) + 1], g.h_status[8 * (g.n_slices - 1) + 2] * 0.01, g.h_status[8 * (g.n_slices - 1) + 3]);
    }
    if (getenv("H264MI_SLICE_TIMELINE")) { // diagnostics (-DMI_ENT_STATS=3 builds): when the slices of each type started and ended within the pass
        Stage &g = d->stage[d->exec];
        uint32_t t0 = 0xFFFFFFFFu;
        for (int i = 0; i < g.n_slices; i++) t0 = std::min(t0, g.h_status[8 * i + 4]);
        for (int ty = 0; ty < 3; ty++) {
            int n = 0;
            double first_end = 1e30, last_start = 0, last_end = 0, sum = 0;
            for (int i = 0; i < g.n_slices; i++)
                if (g.h_slices[i].slice_type % 5 == ty) {
                    const double a = (g.h_status[8 * i + 4] - t0) * 0.01, b = (g.h_status[8 * i + 5] - t0) * 0.01;
                    n++, sum += b - a, first_end = std::min(first_end, b), last_start = std::max(last_start, a), last_end = std::max(last_end, b);
                }
            if (n) fprintf(stderr, "timeline: slice type %d: %d slices, mean %.1f us, first end %.1f, last start %.1f, last end %.1f us\n", ty, n, sum / n, first_end, last_start, last_end);
            if (n > 64) { // slices in 20 ms buckets of their end, split into those that started with the launch and the late ones, with their mean durations
                int cnt[2][16] = {};
                double dur[2][16] = {};
                for (int i = 0; i < g.n_slices; i++)
                    if (g.h_slices[i].slice_type % 5 == ty) {
                        const double a = (g.h_status[8 * i + 4] - t0) * 0.01, b = (g.h_status[8 * i + 5] - t0) * 0.01;
                        const int late = a > 5000.0, k = std::min(15, static_cast<int>(b / 20000.0));
                        cnt[late][k]++, dur[late][k] += b - a;
                    }
                for (int late = 0; late < 2; late++)
                    for (int k = 0; k < 16; k++)
                        if (cnt[late][k]) fprintf(stderr, "timeline:   %s, end in %3d..%3d ms: %5d slices, mean duration %.1f ms\n", late ? "late start" : "first wave", 20 * k, 20 * k + 20, cnt[late][k], dur[late][k] / cnt[late][k] * 1e-3);
            }
        }
    }
#endif
    if (*d->h_xstatus) { // a banded kernel gave up waiting for its neighbour workgroup: the pictures of that launch are wrong
        set_error("reconstruction hand-off timed out (code 0x%08x)", *d->h_xstatus);
        *d->h_xstatus = 0;
        HIP_TRY(hipMemset(d->d_xctl + 64, 0, sizeof(uint32_t)));
        return H264MI_EDEVICE;
    }
    {
        const int r = retry_exhausted(d); // (leaves every stream drained again)
        if (r != H264MI_OK) return r;
    }
    // every batch executed since the last look (pipelined callers synchronise once for several), the most recent one last
    int result = H264MI_OK;
    for (int k = 1; k <= MI_STAGES; k++) {
        const int r = harvest_status(d, d->stage[(d->exec + k) % MI_STAGES]);
        if (r != H264MI_OK) result = r;
    }
    return d->isolate ? H264MI_OK : result;
}

extern "C" int32_t h264mi_decode_batch(h264mi_decoder *d, int32_t n, const uint8_t *const *bufs, const size_t *lens, h264mi_batch_info *info) {
    int r = h264mi_batch_prepare(d, n, bufs, lens, info);
    if (r != H264MI_OK) return r;
    r = h264mi_batch_execute(d);
    if (r != H264MI_OK) return r;
    return h264mi_batch_sync(d);
}

extern "C" int32_t h264mi_last_kernel_times(h264mi_decoder *d, double ms[5]) {
    if (!d || !ms) return H264MI_EINVAL;
    for (int i = 0; i < 5; i++) ms[i] = d->k_ms[i];
    return H264MI_OK;
}

extern "C" int32_t h264mi_last_launch_times(h264mi_decoder *d, int32_t kernel, float *ms, int32_t cap, int32_t *n) {
    if (!d || !n || kernel < 0 || kernel > 3 || cap < 0 || (cap && !ms)) return H264MI_EINVAL;
    const std::vector<float> &v = d->launch_ms[kernel];
    for (int i = 0; i < cap && i < static_cast<int>(v.size()); i++) ms[i] = v[i];
    *n = static_cast<int32_t>(v.size());
    return H264MI_OK;
}

extern "C" int32_t h264mi_stream_frame_count(h264mi_decoder *d, int32_t stream, int32_t *n) {
    if (!d || !n || stream < 0 || stream >= static_cast<int>(d->st.size())) return H264MI_EINVAL;
    *n = static_cast<int32_t>(d->stage[d->exec].out[stream].size());
    return H264MI_OK;
}

static int frame_ptrs(h264mi_decoder *d, int stream, int frame, uint8_t **y, const OutFrame **of) {
    if (!d || stream < 0 || stream >= static_cast<int>(d->st.size())) return H264MI_EINVAL;
    const std::vector<OutFrame> &out = d->stage[d->exec].out[stream];
    if (frame < 0 || frame >= static_cast<int>(out.size())) return H264MI_EINVAL;
    *of = &out[frame];
    *y = d->d_frames + (static_cast<size_t>(stream) * d->n_slots + out[frame].slot) * d->slot_bytes;
    return H264MI_OK;
}

extern "C" int32_t h264mi_frame_device_planes(h264mi_decoder *d, int32_t stream, int32_t frame, void **y, void **cb, void **cr, int32_t *pitch_y, int32_t *pitch_c,
                                              int32_t *cw, int32_t *ch) {
    uint8_t *p;
    const OutFrame *of;
    int r = frame_ptrs(d, stream, frame, &p, &of);
    if (r != H264MI_OK) return r;
    const int W = of->wmb * 16, H = of->hmb * 16;
    if (y) *y = p;
    if (cb) *cb = p + static_cast<size_t>(W) * H;
    if (cr) *cr = p + static_cast<size_t>(W) * H * 5 / 4;
    if (pitch_y) *pitch_y = W;
    if (pitch_c) *pitch_c = W / 2;
    if (cw) *cw = W;
    if (ch) *ch = H;
    return H264MI_OK;
}

extern "C" int32_t h264mi_frame_get_info(h264mi_decoder *d, int32_t stream, int32_t frame, h264mi_frame_info *fi) {
    uint8_t *p;
    const OutFrame *of;
    int r = frame_ptrs(d, stream, frame, &p, &of);
    if (r != H264MI_OK) return r;
    if (!fi) return H264MI_EINVAL;
    fi->width = of->width, fi->height = of->height, fi->coded_width = of->wmb * 16, fi->coded_height = of->hmb * 16;
    fi->crop_x = of->crop_x, fi->crop_y = of->crop_y;
    fi->pic_order_cnt = of->poc, fi->frame_num = of->frame_num, fi->nal_ref_idc = of->nal_ref_idc, fi->idr = of->idr;
    fi->new_sequence = of->new_sequence;
    return H264MI_OK;
}

extern "C" int32_t h264mi_stream_output_order(h264mi_decoder *d, int32_t stream, int32_t *order, int32_t cap, int32_t *n) {
    if (!d || !n || stream < 0 || stream >= static_cast<int>(d->st.size()) || cap < 0 || (cap && !order)) return H264MI_EINVAL;
    const std::vector<OutFrame> &out = d->stage[d->exec].out[stream];
    std::vector<int> idx(out.size()), seq(out.size());
    int cur = 0;
    for (size_t i = 0; i < out.size(); i++) {
        if (i && out[i].new_sequence) cur++;
        idx[i] = static_cast<int>(i), seq[i] = cur;
    }
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return seq[a] != seq[b] ? seq[a] < seq[b] : out[a].poc < out[b].poc; });
    *n = static_cast<int32_t>(out.size());
    for (int i = 0; i < cap && i < static_cast<int>(idx.size()); i++) order[i] = idx[i];
    return H264MI_OK;
}

// the picture's own geometry (a new SPS may have activated later in the same batch)
static void crop_rect(const OutFrame *of, int crop, int *x0, int *y0, int *w, int *h) {
    *x0 = *y0 = 0, *w = of->wmb * 16, *h = of->hmb * 16;
    if (crop) *x0 = of->crop_x, *y0 = of->crop_y, *w = of->width, *h = of->height;
}

extern "C" int32_t h264mi_frame_read(h264mi_decoder *d, int32_t stream, int32_t frame, int32_t crop, uint8_t *dst, size_t cap) {
    uint8_t *p;
    const OutFrame *of;
    int x0, y0, w, h;
    int r = frame_ptrs(d, stream, frame, &p, &of);
    if (r != H264MI_OK) return r;
    if (!dst) return H264MI_EINVAL;
    GUARD(d);
    const int W = of->wmb * 16, H = of->hmb * 16;
    crop_rect(of, crop, &x0, &y0, &w, &h);
    if (cap < static_cast<size_t>(w) * h * 3 / 2) return H264MI_ECAPACITY;
    HIP_TRY(hipStreamSynchronize(d->stream));
    HIP_TRY(hipMemcpy2D(dst, w, p + static_cast<size_t>(y0) * W + x0, W, w, h, hipMemcpyDeviceToHost));
    const uint8_t *cb = p + static_cast<size_t>(W) * H, *cr = cb + static_cast<size_t>(W) * H / 4;
    uint8_t *o = dst + static_cast<size_t>(w) * h;
    HIP_TRY(hipMemcpy2D(o, w / 2, cb + static_cast<size_t>(y0 / 2) * (W / 2) + x0 / 2, W / 2, w / 2, h / 2, hipMemcpyDeviceToHost));
    o += static_cast<size_t>(w / 2) * (h / 2);
    HIP_TRY(hipMemcpy2D(o, w / 2, cr + static_cast<size_t>(y0 / 2) * (W / 2) + x0 / 2, W / 2, w / 2, h / 2, hipMemcpyDeviceToHost));
    return H264MI_OK;
}

// K6 over a list of frames: descriptors go through a small pinned table, one launch packs them all.
static int pack_frames(h264mi_decoder *d, const std::vector<std::pair<int, int>> &frames /* (stream, frame) */, void *dst, size_t cap, size_t *bytes) {
    GUARD(d);
    const int ps = d->pack_slot ^= 1;
    if (!d->ev_pack[ps]) HIP_TRY(hipEventCreateWithFlags(&d->ev_pack[ps], hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(d->ev_pack[ps])); // the launch that used this table (two calls ago) has read it
    if (frames.size() > d->pack_cap[ps]) {
        if (d->h_pack[ps]) hipHostFree(d->h_pack[ps]);
        if (d->d_pack[ps]) hipFree(d->d_pack[ps]);
        d->h_pack[ps] = nullptr, d->d_pack[ps] = nullptr;
        d->pack_cap[ps] = std::max<size_t>(frames.size(), 64);
        HIP_TRY(hipHostMalloc(&d->h_pack[ps], sizeof(PackDesc) * d->pack_cap[ps]));
        HIP_TRY(hipMalloc(&d->d_pack[ps], sizeof(PackDesc) * d->pack_cap[ps]));
    }
    size_t off = 0;
    int hmax = 0;
    for (size_t i = 0; i < frames.size(); i++) {
        uint8_t *p;
        const OutFrame *of;
        int x0, y0, w, h;
        int r = frame_ptrs(d, frames[i].first, frames[i].second, &p, &of);
        if (r != H264MI_OK) return r;
        crop_rect(of, 1, &x0, &y0, &w, &h);
        PackDesc &pk = d->h_pack[ps][i];
        pk.src = reinterpret_cast<uint64_t>(p), pk.dst_off = off;
        pk.W = of->wmb * 16, pk.H = of->hmb * 16, pk.x0 = x0, pk.y0 = y0, pk.w = w, pk.h = h;
        off += static_cast<size_t>(w) * h * 3 / 2;
        hmax = std::max(hmax, h);
    }
    if (bytes) *bytes = off;
    if (off > cap) return H264MI_ECAPACITY;
    if (frames.empty()) return H264MI_OK;
    // (in order on the decoder's stream, which has waited for the reconstruction of the last executed pass: h264mi_batch_execute)
    HIP_TRY(hipMemcpyAsync(d->d_pack[ps], d->h_pack[ps], sizeof(PackDesc) * frames.size(), hipMemcpyHostToDevice, d->stream));
    const int rows_per_block = 32;
    hipLaunchKernelGGL(k_pack, dim3(static_cast<uint32_t>(frames.size()), (2 * hmax + rows_per_block - 1) / rows_per_block), dim3(256), 0, d->stream, d->d_pack[ps],
                       static_cast<uint8_t *>(dst), rows_per_block);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(d->ev_pack[ps], d->stream));
    d->last_pack = d->ev_pack[ps];
    return H264MI_OK;
}

extern "C" int32_t h264mi_frame_pack_device(h264mi_decoder *d, int32_t stream, int32_t frame, void *dst, size_t cap) {
    if (!d || !dst) return H264MI_EINVAL;
    return pack_frames(d, {{stream, frame}}, dst, cap, nullptr);
}

extern "C" int32_t h264mi_batch_pack_device(h264mi_decoder *d, int32_t stream, void *dst, size_t cap, size_t *bytes) {
    if (!d || !dst || stream < -1 || stream >= static_cast<int>(d->st.size())) return H264MI_EINVAL;
    std::vector<std::pair<int, int>> frames;
    const Stage &g = d->stage[d->exec];
    for (int si = 0; si < static_cast<int>(g.out.size()); si++)
        if (stream < 0 || stream == si)
            for (int f = 0; f < static_cast<int>(g.out[si].size()); f++) frames.push_back({si, f});
    return pack_frames(d, frames, dst, cap, bytes);
}

extern "C" int32_t h264mi_frame_read_mbrecs(h264mi_decoder *d, int32_t stream, int32_t frame, uint8_t *rec, size_t cap) {
    if (!d || !rec || stream < 0 || stream >= static_cast<int>(d->st.size())) return H264MI_EINVAL;
    GUARD(d);
    Stage &g = d->stage[d->exec];
    for (int i = 0; i < g.n_pics; i++) {
        const PicDesc &pd = g.h_pics[i];
        if (static_cast<int>(pd.stream) == stream && static_cast<int>(pd.order) == frame) {
            size_t n = static_cast<size_t>(pd.wmb) * pd.hmb * sizeof(MbRec);
            if (cap < n) return H264MI_ECAPACITY;
            HIP_TRY(hipStreamSynchronize(d->stream));
            HIP_TRY(hipMemcpy(rec, d->d_mbrec[(d->pass + MI_SETS - 1) % MI_SETS] + pd.mb_base, n, hipMemcpyDeviceToHost));
            return H264MI_OK;
        }
    }
    return H264MI_EINVAL;
}

extern "C" int32_t h264mi_frame_read_mbmv1(h264mi_decoder *d, int32_t stream, int32_t frame, uint8_t *mv1, size_t cap) {
    if (!d || !mv1 || stream < 0 || stream >= static_cast<int>(d->st.size())) return H264MI_EINVAL;
    GUARD(d);
    Stage &g = d->stage[d->exec];
    const int set = static_cast<int>((d->pass + MI_SETS - 1) % MI_SETS);
    for (int i = 0; i < g.n_pics; i++) {
        const PicDesc &pd = g.h_pics[i];
        if (static_cast<int>(pd.stream) == stream && static_cast<int>(pd.order) == frame) {
            size_t n = static_cast<size_t>(pd.wmb) * pd.hmb * sizeof(MbMv1);
            if (cap < n) return H264MI_ECAPACITY;
            HIP_TRY(hipStreamSynchronize(d->stream));
            if (!pd.has_b || !d->d_mv1[set]) { // no B slice in the picture: there are no list-1 vectors
                memset(mv1, 0, n);
                return H264MI_OK;
            }
            HIP_TRY(hipMemcpy(mv1, d->d_mv1[set] + pd.mb_base, n, hipMemcpyDeviceToHost));
            return H264MI_OK;
        }
    }
    return H264MI_EINVAL;
}

#if defined(H264MI_TEST_HOOKS)
// ---- test and measurement hooks: built into libh264mi_hooks.so only (csrc/Makefile); the product library exports the ABI of include/h264mi.h and nothing else ----
// Not part of the public ABI: fills every intermediate buffer (macroblock records, coefficient blocks, row state) with
// 0xFF so that a test can prove that no kernel depends on what an earlier batch -- or the allocator -- left there.
extern "C" int32_t h264mi_internal_poison(h264mi_decoder *d) {
    if (!d) return H264MI_EINVAL;
    GUARD(d);
    for (int i = 0; i < 2; i++) HIP_TRY(hipStreamSynchronize(d->ent_stream[i]));
    HIP_TRY(hipStreamSynchronize(d->rec_stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    for (int i = 0; i < MI_SETS; i++) {
        HIP_TRY(hipMemset(d->d_mbrec[i], 0xFF, sizeof(MbRec) * d->mb_cap));
        if (i == 0) HIP_TRY(hipMemset(d->d_coef[0], 0xFF, d->pool_blocks * 32 * MI_SETS));
        HIP_TRY(hipMemset(d->d_toprows[i], 0xFF, static_cast<size_t>(d->slices_cap) * (d->Wmax / 16) * MI_TOPROW_BYTES));
        if (d->d_mv1[i]) HIP_TRY(hipMemset(d->d_mv1[i], 0xFF, sizeof(MbMv1) * d->mb_cap));
    }
    if (d->d_colrec) HIP_TRY(hipMemset(d->d_colrec, 0xFF, sizeof(ColRec) * d->colrec_per_slot * d->n_slots * d->st.size()));
    HIP_TRY(hipDeviceSynchronize());
    return H264MI_OK;
}

// Not part of the public ABI: lets the CPU test-suite check the K5 launch plan (wavefronts, hand-off ring depth,
// dynamic LDS) without a GPU -- a wrong plan would deadlock the kernel, see mi_deblock8_plan().
extern "C" int32_t h264mi_internal_deblock_plan(int32_t wmb, int32_t hmb, int32_t *nwaves, int32_t *ring, int32_t *ring_last, int32_t *last_bufs, int64_t *lds_bytes) {
    if (!nwaves || !ring || !ring_last || !last_bufs || !lds_bytes || wmb < 1 || hmb < 1) return H264MI_EINVAL;
    int w = 1, r = 1, rl = 1, nb = 1;
    mi_deblock8_plan(wmb, hmb, &w, &r, &rl, &nb);
    *nwaves = w, *ring = r, *ring_last = rl, *last_bufs = nb, *lds_bytes = static_cast<int64_t>(mi_deblock8_lds_bytes(w, r, rl, nb));
    return H264MI_OK;
}


// Not part of the public ABI: the phase clocks a -DMI_DB_STATS build of the banded deblocking kernels leaves (zero otherwise); reading clears them
extern "C" int32_t h264mi_internal_deblock_phase_clocks(h264mi_decoder *d, uint32_t out[12]) {
    if (!d || !out) return H264MI_EINVAL;
    GUARD(d);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d->d_xctl + 64 + 8, 12 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(d->d_xctl + 64 + 8, 0, 12 * sizeof(uint32_t)));
    return H264MI_OK;
}

// Not part of the public ABI: start and end (100 MHz ticks) of the row groups of the first picture of the last K5 launch (-DMI_DB_STATS builds)
extern "C" int32_t h264mi_internal_deblock_group_times(h264mi_decoder *d, uint32_t out[128]) { // [0..31] start / end per group, [32 + 8 g + i] when group g began step 16 i
    if (!d || !out) return H264MI_EINVAL;
    GUARD(d);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d->d_xctl + 64 + 32, 128 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return H264MI_OK;
}

// Not part of the public ABI: the banded launch plan of K5 / K3 (mi_deblock_bands, mi_intra_bands) for the CPU model test.
extern "C" int32_t h264mi_internal_band_plan(int32_t n_pics, int32_t wmb, int32_t hmb, int32_t max_wgs, int32_t *k5_bands, int32_t *k5_waves, int32_t *k5_ring,
                                             int64_t *k5_lds, int32_t *k3_bands, int32_t *k3_waves) {
    if (!k5_bands || !k5_waves || !k5_ring || !k5_lds || !k3_bands || !k3_waves || n_pics < 1 || wmb < 1 || hmb < 1) return H264MI_EINVAL;
    int nb = 1, nw = 1, ring = 1, b3 = 1, w3 = 1, roles = 1;
    mi_deblock_bands(n_pics, wmb, hmb, max_wgs, &nb, &nw, &ring, &roles);
    int wpr = 1;
    mi_intra_bands(n_pics, hmb, max_wgs, false, &b3, &w3, &wpr);
    // (the model test works on groups: the kernel runs `roles` wavefronts on each, which follow the same protocol side by side)
    *k5_bands = nb, *k5_waves = nw / roles, *k5_ring = ring, *k5_lds = static_cast<int64_t>(mi_deblock_lds_bytes_banded(nw, ring));
    *k3_bands = b3, *k3_waves = w3;
    return H264MI_OK;
}

// Not part of the public ABI: the epoch / ticket counters of the banded kernels set to a value shortly before their 32-bit wrap
// (tests/test_gpu_parity.py::test_gpu_banded_kernels_across_the_epoch_wrap)
extern "C" int32_t h264mi_internal_set_epoch(h264mi_decoder *d, uint32_t v) {
    if (!d) return H264MI_EINVAL;
    GUARD(d);
    for (int i = 0; i < 2; i++) HIP_TRY(hipStreamSynchronize(d->ent_stream[i]));
    HIP_TRY(hipStreamSynchronize(d->rec_stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    d->x_epoch = d->x_tk5 = d->x_tk3 = v;
    // the device-side ticket counters stand where the host's bases do
    HIP_TRY(hipMemcpy(d->d_xctl, &d->x_tk5, sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d->d_xctl + 32, &d->x_tk3, sizeof(uint32_t), hipMemcpyHostToDevice));
    return H264MI_OK;
}
#endif /* H264MI_TEST_HOOKS */

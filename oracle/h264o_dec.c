/*
 * oracle/h264o_dec.c -- stream driver: NAL dispatch (h264/server.go:113-166), picture
 * boundaries (7.4.1.2.4), POC (8.2.1), reference lists (8.2.4), reference marking (8.2.5),
 * frame output with cropping.
 *
 * TEST INFRASTRUCTURE ONLY (see h264o.h).
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "h264o_int.h"

int h264o_fail(h264o_decoder *d, const char *fmt, ...) {
    if (!d->info.error) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(d->err, sizeof(d->err), fmt, ap);
        va_end(ap);
        d->info.error = 1;
    }
    return -1;
}
const char *h264o_last_error(h264o_decoder *d) { return d->err; }

h264o_decoder *h264o_decoder_create(void) { return (h264o_decoder *)calloc(1, sizeof(h264o_decoder)); }
static void free_pics(h264o_decoder *d) {
    for (int i = 0; i < d->n_pics; i++) free(d->pics[i].plane[0]), free(d->pics[i].mbs), free(d->fviews[i][0].mbs), free(d->fviews[i][1].mbs);
    memset(d->fviews, 0, sizeof(d->fviews));
    d->n_pics = 0;
    free(d->mb);
    d->mb = NULL;
}
void h264o_decoder_destroy(h264o_decoder *d) {
    if (!d) return;
    free_pics(d);
    free(d->rbsp);
    free(d->sgmap);
    for (int i = 0; i < 256; i++) free(d->sg_ids[i]);
    free(d);
}
void h264o_set_mb_trace(h264o_decoder *d, int32_t *trace, size_t cap) {
    d->trace = trace;
    d->trace_cap = cap;
    d->trace_pos = 0;
}

static int activate(h264o_decoder *d, const h264o_pps *pps) {
    const h264o_sps *s = &d->sps[pps->seq_parameter_set_id];
    if (!s->valid) return h264o_fail(d, "PPS %d refers to missing SPS %d", pps->pic_parameter_set_id, pps->seq_parameter_set_id);
    /* frame_mbs_only_flag = 0 is accepted as long as every picture is a frame and macroblock-adaptive coding is off (h264/sps.go:316-322,
     * h264/slice.go:867-872): such pictures decode like progressive ones, only the map units are two macroblock rows high */
    /* chroma_format_idc 0 (monochrome, h264/sps.go:226-243): decoded as 4:2:0 whose chroma planes are 128 -- nothing of them is in the stream, their intra
     * prediction is the DC of planes that are 128 everywhere, their residual is zero */
    if (s->chroma_format_idc > 1 || s->bit_depth_luma_minus8 || s->bit_depth_chroma_minus8 || (!s->frame_mbs_only_flag && s->mb_adaptive_frame_field_flag) ||
        s->qpprime_y_zero_transform_bypass_flag)
        return h264o_fail(d, "unsupported SPS (need 4:2:0 or monochrome, 8-bit, no MBAFF; chroma_format_idc=%d)", s->chroma_format_idc);
    int wmb = s->pic_width_in_mbs_minus1 + 1, hmb = (s->pic_height_in_map_units_minus1 + 1) * (2 - s->frame_mbs_only_flag); /* h264/slice.go:159-176 */
    if (d->asps != s || wmb != d->wmb || hmb != d->fhmb || !d->mb) {
        free_pics(d);
        d->wmb = wmb;
        d->hmb = d->fhmb = hmb;
        d->pend = NULL;
        d->mb = (h264o_mb *)calloc((size_t)wmb * hmb, sizeof(h264o_mb));
        int n = (s->max_num_ref_frames > 0 ? s->max_num_ref_frames : 1) + 2;
        if (n > 20) n = 20;
        size_t ysz = (size_t)wmb * 16 * hmb * 16;
        for (int i = 0; i < n; i++) {
            h264o_pic *p = &d->pics[i];
            memset(p, 0, sizeof(*p));
            p->plane[0] = (uint8_t *)malloc(ysz * 3 / 2);
            p->plane[1] = p->plane[0] + ysz;
            p->plane[2] = p->plane[1] + ysz / 4;
            p->stride[0] = wmb * 16;
            p->stride[1] = p->stride[2] = wmb * 8;
            p->parity = -1;
        }
        d->n_pics = n;
    }
    d->info.coded_width = wmb * 16;
    d->info.coded_height = hmb * 16;
    d->info.width = wmb * 16 - 2 * (s->frame_crop_left_offset + s->frame_crop_right_offset);
    d->info.height = hmb * 16 - 2 * (2 - s->frame_mbs_only_flag) * (s->frame_crop_top_offset + s->frame_crop_bottom_offset); /* CropUnitY = SubHeightC * (2 - frame_mbs_only_flag) */
    d->asps = s;
    d->apps = pps;
    h264o_build_level_scale(d);
    return 0;
}

/* ------------------------------------------------------------------ 8.2.1 picture order count */
static int compute_poc(h264o_decoder *d, const h264o_slice_header *sh) {
    const h264o_sps *s = d->asps;
    int max_fn = 1 << (s->log2_max_frame_num_minus4 + 4);
    int poc = 0;
    if (s->pic_order_cnt_type == 0) {
        int max_lsb = 1 << (s->log2_max_pic_order_cnt_lsb_minus4 + 4);
        int prev_msb = sh->idr_flag ? 0 : d->prev_poc_msb, prev_lsb = sh->idr_flag ? 0 : d->prev_poc_lsb, msb;
        if (sh->pic_order_cnt_lsb < prev_lsb && prev_lsb - sh->pic_order_cnt_lsb >= max_lsb / 2)
            msb = prev_msb + max_lsb;
        else if (sh->pic_order_cnt_lsb > prev_lsb && sh->pic_order_cnt_lsb - prev_lsb > max_lsb / 2)
            msb = prev_msb - max_lsb;
        else
            msb = prev_msb;
        int top = msb + sh->pic_order_cnt_lsb, bot = top + sh->delta_pic_order_cnt_bottom;
        poc = sh->field_pic_flag ? top /* 8-4 / 8-5: a field has the one count */ : (top < bot ? top : bot);
        d->poc_top = top, d->poc_bot = sh->field_pic_flag ? top : bot;
        if (sh->nal_ref_idc) {
            d->prev_poc_msb = msb;
            d->prev_poc_lsb = sh->pic_order_cnt_lsb;
        }
    } else {
        int fno;
        if (sh->idr_flag)
            fno = 0;
        else if (d->prev_frame_num > sh->frame_num)
            fno = d->prev_frame_num_offset + max_fn;
        else
            fno = d->prev_frame_num_offset;
        if (s->pic_order_cnt_type == 1) {
            int abs_fn = s->num_ref_frames_in_pic_order_cnt_cycle ? fno + sh->frame_num : 0;
            if (!sh->nal_ref_idc && abs_fn > 0) abs_fn--;
            int expected = 0;
            if (abs_fn > 0) {
                int cyc = (abs_fn - 1) / s->num_ref_frames_in_pic_order_cnt_cycle, in_cyc = (abs_fn - 1) % s->num_ref_frames_in_pic_order_cnt_cycle;
                int delta_cycle = 0;
                for (int i = 0; i < s->num_ref_frames_in_pic_order_cnt_cycle; i++) delta_cycle += s->offset_for_ref_frame[i];
                expected = cyc * delta_cycle;
                for (int i = 0; i <= in_cyc; i++) expected += s->offset_for_ref_frame[i];
            }
            if (!sh->nal_ref_idc) expected += s->offset_for_non_ref_pic;
            int top = expected + sh->delta_pic_order_cnt[0], bot = top + s->offset_for_top_to_bottom_field + sh->delta_pic_order_cnt[1];
            if (sh->field_pic_flag) /* 8-10: a bottom field is expected + offset_for_top_to_bottom_field + delta_pic_order_cnt[0] */
                poc = sh->bottom_field_flag ? expected + s->offset_for_top_to_bottom_field + sh->delta_pic_order_cnt[0] : top;
            else
                poc = top < bot ? top : bot;
            d->poc_top = top, d->poc_bot = sh->field_pic_flag ? poc : bot;
        } else
            poc = sh->idr_flag ? 0 : (sh->nal_ref_idc ? 2 * (fno + sh->frame_num) : 2 * (fno + sh->frame_num) - 1);
        if (s->pic_order_cnt_type == 2) d->poc_top = d->poc_bot = poc;
        d->prev_frame_num_offset = fno;
    }
    d->prev_frame_num = sh->frame_num;
    return poc;
}

/* ------------------------------------------------------------------ 8.2.4 reference picture lists */
static int cmp_picnum_desc(const void *a, const void *b) { return (*(h264o_pic *const *)b)->pic_num - (*(h264o_pic *const *)a)->pic_num; }
static int cmp_ltidx_asc(const void *a, const void *b) {
    return (*(h264o_pic *const *)a)->long_term_frame_idx - (*(h264o_pic *const *)b)->long_term_frame_idx;
}
static int cmp_poc_desc(const void *a, const void *b) { return (*(h264o_pic *const *)b)->poc - (*(h264o_pic *const *)a)->poc; }
static int cmp_poc_asc(const void *a, const void *b) { return (*(h264o_pic *const *)a)->poc - (*(h264o_pic *const *)b)->poc; }

/* 8.2.4.3 modification of one list (h264/slice.go:909-936 parses the commands) */
static int modify_list(h264o_decoder *d, h264o_pic **list, int nact, h264o_pic **st, int nst, h264o_pic **lt, int nlt, int ncmd, const int *idc, const int *val) {
    const h264o_slice_header *sh = &d->sh;
    int max_fn = 1 << (d->asps->log2_max_frame_num_minus4 + 4);
    int pred = sh->frame_num, idx = 0; /* picNumLXPred = CurrPicNum */
    for (int k = 0; k < ncmd && idx < nact; k++) {
        h264o_pic *target = NULL;
        if (idc[k] < 2) {
            int diff = val[k] + 1;
            if (idc[k] == 0) {
                pred -= diff;
                if (pred < 0) pred += max_fn;
            } else {
                pred += diff;
                if (pred >= max_fn) pred -= max_fn;
            }
            int picnum = pred > sh->frame_num ? pred - max_fn : pred;
            for (int i = 0; i < nst; i++)
                if (st[i]->pic_num == picnum) target = st[i];
        } else {
            for (int i = 0; i < nlt; i++)
                if (lt[i]->long_term_frame_idx == val[k]) target = lt[i];
        }
        if (!target) return h264o_fail(d, "ref_pic_list_modification names a missing picture");
        d->feat |= 1u << (8 + idc[k]);
        /* shift up, insert, remove the duplicate further down (8-37/8-38) */
        for (int c = nact; c > idx; c--) list[c] = list[c - 1];
        list[idx++] = target;
        int nidx = idx;
        for (int c = idx; c <= nact; c++)
            if (list[c] != target) list[nidx++] = list[c];
    }
    return 0;
}

/* 8.2.4.3 for field pictures: the commands name FIELDS -- picNumF of a short-term field is 2 * FrameNumWrap + 1 for a field of the
 * current parity and 2 * FrameNumWrap for one of the other, LongTermPicNum likewise on LongTermFrameIdx; CurrPicNum = 2 * frame_num + 1,
 * MaxPicNum = 2 * MaxFrameNum (8.2.4.1).  st / lt: the reference frames (the current one included for a second field). */
static int modify_field_list(h264o_decoder *d, h264o_pic **list, int nact, h264o_pic **st, int nst, h264o_pic **lt, int nlt, int ncmd, const int *idc, const int *val) {
    const h264o_slice_header *sh = &d->sh;
    const int max_pic_num = 2 << (d->asps->log2_max_frame_num_minus4 + 4), cur_pic_num = 2 * sh->frame_num + 1;
    int pred = cur_pic_num, idx = 0;
    for (int k = 0; k < ncmd && idx < nact; k++) {
        h264o_pic *target = NULL;
        if (idc[k] < 2) {
            const int diff = val[k] + 1;
            if (idc[k] == 0) {
                pred -= diff;
                if (pred < 0) pred += max_pic_num;
            } else {
                pred += diff;
                if (pred >= max_pic_num) pred -= max_pic_num;
            }
            const int picnum = pred > cur_pic_num ? pred - max_pic_num : pred;
            for (int i = 0; i < nst; i++)
                for (int par = 0; par < 2; par++)
                    if (((st[i]->fields & ~st[i]->funref) >> par & 1) && 2 * st[i]->frame_num_wrap + (par == d->bottom) == picnum) target = &d->fviews[st[i] - d->pics][par];
        } else {
            for (int i = 0; i < nlt; i++)
                for (int par = 0; par < 2; par++)
                    if (((lt[i]->fields & ~lt[i]->funref) >> par & 1) && 2 * lt[i]->long_term_frame_idx + (par == d->bottom) == val[k]) target = &d->fviews[lt[i] - d->pics][par];
        }
        if (!target) return h264o_fail(d, "ref_pic_list_modification names a missing field");
        d->feat |= 1u << (8 + idc[k]);
        for (int c = nact; c > idx; c--) list[c] = list[c - 1];
        list[idx++] = target;
        int nidx = idx;
        for (int c = idx; c <= nact; c++)
            if (list[c] != target) list[nidx++] = list[c];
    }
    return 0;
}

static int build_ref_list(h264o_decoder *d) {
    const h264o_slice_header *sh = &d->sh;
    int max_fn = 1 << (d->asps->log2_max_frame_num_minus4 + 4);
    h264o_pic *st[20], *lt[20];
    int nst = 0, nlt = 0;
    for (int i = 0; i < d->n_pics; i++) {
        h264o_pic *p = &d->pics[i];
        if (p == d->curf && !(d->field_pic && d->second_field && p->ref == 1)) continue; /* (a second field may predict from the first field of its frame) */
        if (!d->field_pic && p->ref && (p->funref || p->fields != 3)) continue; /* 8.2.4.2.1: a frame picture predicts from frames of which both fields are there and marked */
        if (p->ref == 1) {
            p->frame_num_wrap = p->frame_num > sh->frame_num ? p->frame_num - max_fn : p->frame_num;
            p->pic_num = p->frame_num_wrap;
            st[nst++] = p;
        } else if (p->ref == 2)
            lt[nlt++] = p;
    }
    qsort(lt, nlt, sizeof(lt[0]), cmp_ltidx_asc);
    int nact0 = sh->num_ref_idx_l0_active_minus1 + 1, nact1 = sh->num_ref_idx_l1_active_minus1 + 1;
    memset(d->rpl0, 0, sizeof(d->rpl0));
    memset(d->rpl1, 0, sizeof(d->rpl1));
    if (nst + nlt == 0) return h264o_fail(d, "P/B slice without reference pictures");
    int n0 = 0, n1 = 0;
    if (d->field_pic) {
        /* 8.2.4.2.2 / 8.2.4.2.4 + 8.2.4.2.5: the reference frames in order -- P: by FrameNumWrap; B: by PicOrderCnt around the current field,
         * list 0 the earlier ones first, list 1 the later ones; long-term: by LongTermFrameIdx --, then their fields alternately, the parity
         * of the current field first; a missing field is passed over, and when one parity is used up the other one follows in order */
        h264o_pic *ord[2][20];
        int nord[2] = {0, 0};
        if (sh->slice_type != 1) {
            for (int i = 0; i < nst; i++)
                for (int j = i + 1; j < nst; j++)
                    if (st[j]->frame_num_wrap > st[i]->frame_num_wrap) {
                        h264o_pic *t = st[i];
                        st[i] = st[j], st[j] = t;
                    }
            for (int i = 0; i < nst; i++) ord[0][nord[0]++] = st[i];
        } else {
            /* PicOrderCnt of a frame of which both fields are references: the smaller one; of the current frame (second field): its first field's */
            h264o_pic *before[20], *after[20];
            int nb = 0, na = 0;
            const int cur_poc = d->cur->poc;
            for (int i = 0; i < nst; i++) {
                if (st[i]->nonexisting) continue;
                const int fp = st[i] == d->curf ? st[i]->fpoc[!d->bottom] : st[i]->poc;
                st[i]->pic_num = fp; /* (scratch: the sort key) */
                if (fp <= cur_poc) before[nb++] = st[i];
                else after[na++] = st[i];
            }
            for (int i = 0; i < nb; i++)
                for (int j = i + 1; j < nb; j++)
                    if (before[j]->pic_num > before[i]->pic_num) {
                        h264o_pic *t = before[i];
                        before[i] = before[j], before[j] = t;
                    }
            for (int i = 0; i < na; i++)
                for (int j = i + 1; j < na; j++)
                    if (after[j]->pic_num < after[i]->pic_num) {
                        h264o_pic *t = after[i];
                        after[i] = after[j], after[j] = t;
                    }
            for (int i = 0; i < nb; i++) ord[0][nord[0]++] = before[i];
            for (int i = 0; i < na; i++) ord[0][nord[0]++] = after[i];
            for (int i = 0; i < na; i++) ord[1][nord[1]++] = after[i];
            for (int i = 0; i < nb; i++) ord[1][nord[1]++] = before[i];
        }
        for (int l = 0; l < (sh->slice_type == 1 ? 2 : 1); l++) {
            h264o_pic **out = l ? d->rpl1 : d->rpl0;
            int n = 0;
            for (int grp = 0; grp < 2; grp++) {
                h264o_pic **fr = grp ? lt : ord[l];
                const int nfr = grp ? nlt : nord[l];
                int a = 0, b = 0; /* next frame to look at for the same / the opposite parity */
                for (int want_same = 1;; want_same ^= 1) {
                    int *cursor = want_same ? &a : &b;
                    const int par = want_same ? d->bottom : !d->bottom;
                    while (*cursor < nfr && !((fr[*cursor]->fields & ~fr[*cursor]->funref) >> par & 1)) (*cursor)++;
                    if (*cursor == nfr) { /* this parity is used up: the rest of the other one */
                        cursor = want_same ? &b : &a;
                        const int opar = !par;
                        for (; *cursor < nfr; (*cursor)++)
                            if (((fr[*cursor]->fields & ~fr[*cursor]->funref) >> opar & 1) && n < 32) out[n++] = &d->fviews[fr[*cursor] - d->pics][opar];
                        break;
                    }
                    if (n < 32) out[n++] = &d->fviews[fr[*cursor] - d->pics][par];
                    (*cursor)++;
                }
            }
            if (l) n1 = n;
            else n0 = n;
        }
        if (sh->slice_type == 1 && n1 > 1 && n0 == n1 && memcmp(d->rpl0, d->rpl1, sizeof(d->rpl0[0]) * n1) == 0) {
            h264o_pic *t = d->rpl1[0];
            d->rpl1[0] = d->rpl1[1], d->rpl1[1] = t;
        }
        for (int i = nact0; i < 33; i++) d->rpl0[i] = NULL;
        for (int i = nact1; i < 33; i++) d->rpl1[i] = NULL;
        if (sh->ref_pic_list_modification_flag_l0 && modify_field_list(d, d->rpl0, nact0, st, nst, lt, nlt, sh->n_rplm, sh->rplm_idc, sh->rplm_val) < 0) return -1;
        if (sh->slice_type == 1 && sh->ref_pic_list_modification_flag_l1 &&
            modify_field_list(d, d->rpl1, nact1, st, nst, lt, nlt, sh->n_rplm1, sh->rplm1_idc, sh->rplm1_val) < 0)
            return -1;
        for (int i = nact0; i < 33; i++) d->rpl0[i] = NULL;
        for (int i = nact1; i < 33; i++) d->rpl1[i] = NULL;
        return 0;
    }
    if (sh->slice_type != 1) { /* 8.2.4.2.1: P / SP -- PicNum descending, then LongTermPicNum ascending */
        qsort(st, nst, sizeof(st[0]), cmp_picnum_desc);
        for (int i = 0; i < nst && n0 < 32; i++) d->rpl0[n0++] = st[i];
    } else { /* 8.2.4.2.3: B -- by PicOrderCnt relative to the current picture */
        h264o_pic *before[20], *after[20];
        int nb = 0, na = 0, cur_poc = d->cur->poc;
        for (int i = 0; i < nst; i++) {
            if (st[i]->nonexisting && d->asps->pic_order_cnt_type == 0) continue; /* 8.2.4.2.3: no PicOrderCnt, not in the lists of B slices */
            if (st[i]->poc < cur_poc)
                before[nb++] = st[i];
            else
                after[na++] = st[i];
        }
        qsort(before, nb, sizeof(before[0]), cmp_poc_desc);
        qsort(after, na, sizeof(after[0]), cmp_poc_asc);
        for (int i = 0; i < nb; i++) d->rpl0[n0++] = before[i];
        for (int i = 0; i < na; i++) d->rpl0[n0++] = after[i];
        for (int i = 0; i < na; i++) d->rpl1[n1++] = after[i];
        for (int i = 0; i < nb; i++) d->rpl1[n1++] = before[i];
        for (int i = 0; i < nlt && n1 < 32; i++) d->rpl1[n1++] = lt[i];
    }
    for (int i = 0; i < nlt && n0 < 32; i++) d->rpl0[n0++] = lt[i];
    if (sh->slice_type == 1 && n1 > 1 && n0 == n1 && memcmp(d->rpl0, d->rpl1, sizeof(d->rpl0[0]) * n1) == 0) { /* identical lists: swap the first two of list 1 */
        h264o_pic *t = d->rpl1[0];
        d->rpl1[0] = d->rpl1[1], d->rpl1[1] = t;
    }
    /* the initial lists are cut to the active size before modification (8.2.4.2) */
    for (int i = nact0; i < 33; i++) d->rpl0[i] = NULL;
    for (int i = nact1; i < 33; i++) d->rpl1[i] = NULL;
    /* PicNum of short-term pictures for the modification commands; long-term: LongTermPicNum == LongTermFrameIdx for frames */
    if (sh->ref_pic_list_modification_flag_l0 && modify_list(d, d->rpl0, nact0, st, nst, lt, nlt, sh->n_rplm, sh->rplm_idc, sh->rplm_val) < 0) return -1;
    if (sh->slice_type == 1 && sh->ref_pic_list_modification_flag_l1 &&
        modify_list(d, d->rpl1, nact1, st, nst, lt, nlt, sh->n_rplm1, sh->rplm1_idc, sh->rplm1_val) < 0)
        return -1;
    for (int i = nact0; i < 33; i++) d->rpl0[i] = NULL;
    for (int i = nact1; i < 33; i++) d->rpl1[i] = NULL;
    for (int i = 0; i < nact0; i++)
        if (d->rpl0[i] && d->rpl0[i]->ref == 2) d->feat |= 1u << 11;
    /* entries beyond the initial list that were never filled stay NULL; prediction from them is an error */
    return 0;
}

/* ------------------------------------------------------------------ 8.2.5 decoded reference picture marking */
static void mark_reference(h264o_decoder *d) {
    const h264o_slice_header *sh = &d->first_sh;
    h264o_pic *cur = d->curf;
    int max_fn = 1 << (d->asps->log2_max_frame_num_minus4 + 4);
    if (sh->nal_ref_idc) d->prev_ref_frame_num = sh->frame_num;
    if (d->field_pic && sh->nal_ref_idc && !sh->idr_flag && sh->adaptive_ref_pic_marking_mode_flag) {
        /* 8.2.5.4.1 in a field picture: picNumX names a FIELD (8.2.4.1); the frame stays a reference frame while its other field is one */
        const int cur_pic_num = 2 * sh->frame_num + 1;
        for (int k = 0; k < sh->n_mmco; k++) {
            if (sh->mmco_op[k] != 1) {
                h264o_fail(d, "memory_management_control_operation %d in a field picture is out of scope", sh->mmco_op[k]);
                return;
            }
            d->feat |= 1u << 1;
            const int picnum = cur_pic_num - (sh->mmco_arg1[k] + 1);
            for (int i = 0; i < d->n_pics; i++) {
                h264o_pic *p = &d->pics[i];
                if (p->ref != 1) continue;
                const int wrap = p->frame_num > sh->frame_num ? p->frame_num - max_fn : p->frame_num;
                for (int par = 0; par < 2; par++)
                    if (((p->fields & ~p->funref) >> par & 1) && 2 * wrap + (par == d->bottom) == picnum) {
                        p->funref |= 1 << par;
                        if (!(p->fields & ~p->funref)) p->ref = 0; /* (the current frame, whose other field is being decoded, is marked just below) */
                    }
            }
        }
        cur->ref = 1;
        return;
    }
    /* 8.2.5.3: the second field of a frame whose first field is a short-term reference joins it, nothing leaves the window */
    if (d->field_pic && d->second_field && cur->ref) return; /* 7.4.3 (operation 5 below: 0) */
    if (!sh->nal_ref_idc) {
        cur->ref = 0;
        d->feat |= 1u << 12;
        return;
    }
    if (sh->idr_flag) {
        for (int i = 0; i < d->n_pics; i++)
            if (&d->pics[i] != cur) d->pics[i].ref = 0;
        cur->ref = sh->long_term_reference_flag ? 2 : 1;
        cur->long_term_frame_idx = 0;
        return;
    }
    cur->ref = 1;
    if (sh->adaptive_ref_pic_marking_mode_flag) {
        for (int k = 0; k < sh->n_mmco; k++) {
            int op = sh->mmco_op[k];
            d->feat |= 1u << op;
            for (int i = 0; i < d->n_pics; i++) { /* refresh PicNum */
                h264o_pic *p = &d->pics[i];
                if (p->ref == 1) p->pic_num = p->frame_num > sh->frame_num ? p->frame_num - max_fn : p->frame_num;
            }
            if (op == 1 || op == 3) {
                int picnum = sh->frame_num - (sh->mmco_arg1[k] + 1);
                for (int i = 0; i < d->n_pics; i++) {
                    h264o_pic *p = &d->pics[i];
                    if (p != cur && p->ref == 1 && p->pic_num == picnum) {
                        if (op == 1)
                            p->ref = 0;
                        else {
                            for (int j = 0; j < d->n_pics; j++)
                                if (d->pics[j].ref == 2 && d->pics[j].long_term_frame_idx == sh->mmco_arg2[k]) d->pics[j].ref = 0;
                            p->ref = 2;
                            p->long_term_frame_idx = sh->mmco_arg2[k];
                        }
                    }
                }
            } else if (op == 2) {
                for (int i = 0; i < d->n_pics; i++)
                    if (d->pics[i].ref == 2 && d->pics[i].long_term_frame_idx == sh->mmco_arg1[k]) d->pics[i].ref = 0;
            } else if (op == 4) {
                for (int i = 0; i < d->n_pics; i++)
                    if (d->pics[i].ref == 2 && d->pics[i].long_term_frame_idx >= sh->mmco_arg1[k]) d->pics[i].ref = 0;
            } else if (op == 5) {
                for (int i = 0; i < d->n_pics; i++)
                    if (&d->pics[i] != cur) d->pics[i].ref = 0;
                cur->frame_num = 0;
                cur->poc = 0; /* 8.2.1: tempPicOrderCnt is subtracted after decoding */
                d->prev_frame_num = 0;
                d->prev_frame_num_offset = 0;
                d->prev_ref_frame_num = 0;
                d->prev_poc_msb = 0;
                /* 8.2.1.1: prevPicOrderCntLsb = TopFieldOrderCnt after tempPicOrderCnt was subtracted: 0 unless the bottom field is the earlier one */
                d->prev_poc_lsb = d->asps->pic_order_cnt_type == 0 && !d->field_pic && d->poc_bot < d->poc_top ? d->poc_top - d->poc_bot : 0;
            } else if (op == 6) {
                for (int j = 0; j < d->n_pics; j++)
                    if (d->pics[j].ref == 2 && d->pics[j].long_term_frame_idx == sh->mmco_arg2[k]) d->pics[j].ref = 0;
                cur->ref = 2;
                cur->long_term_frame_idx = sh->mmco_arg2[k];
            }
        }
    } else {
        /* 8.2.5.3 sliding window */
        int nref = 0, maxref = d->asps->max_num_ref_frames > 0 ? d->asps->max_num_ref_frames : 1;
        h264o_pic *oldest = NULL;
        for (int i = 0; i < d->n_pics; i++) {
            h264o_pic *p = &d->pics[i];
            if (p == cur || !p->ref) continue;
            nref++;
            if (p->ref == 1) {
                int wrap = p->frame_num > sh->frame_num ? p->frame_num - max_fn : p->frame_num;
                if (!oldest || wrap < oldest->frame_num_wrap) {
                    p->frame_num_wrap = wrap;
                    oldest = p;
                }
            }
        }
        if (nref >= maxref && oldest) oldest->ref = 0;
    }
}

static h264o_pic *make_view(h264o_decoder *d, h264o_pic *p, int parity, int poc);

/* ------------------------------------------------------------------ picture output */
static void emit_frame(h264o_decoder *d, h264o_pic *p) {
    const h264o_sps *s = d->asps;
    int cw = d->info.coded_width, ch = d->info.coded_height;
    int w = d->crop ? d->info.width : cw, h = d->crop ? d->info.height : ch;
    int x0 = d->crop ? 2 * s->frame_crop_left_offset : 0, y0 = d->crop ? 2 * (2 - s->frame_mbs_only_flag) * s->frame_crop_top_offset : 0;
    size_t need = (size_t)w * h * 3 / 2;
    if (d->out && d->out_pos + need <= d->out_cap) {
        uint8_t *o = d->out + d->out_pos;
        for (int y = 0; y < h; y++) memcpy(o + (size_t)y * w, p->plane[0] + (size_t)(y + y0) * p->stride[0] + x0, w);
        o += (size_t)w * h;
        for (int pl = 1; pl < 3; pl++) {
            for (int y = 0; y < h / 2; y++) memcpy(o + (size_t)y * (w / 2), p->plane[pl] + (size_t)(y + y0 / 2) * p->stride[pl] + x0 / 2, w / 2);
            o += (size_t)(w / 2) * (h / 2);
        }
    }
    d->out_pos += need;
    if (d->n_pocs < 8192) d->pocs[d->n_pocs++] = p->poc;
    d->info.n_frames++;
}

static void finish_picture(h264o_decoder *d) {
    if (!d->cur) return;
    /* conceal MBs never covered by a slice (not expected in scope): copy nothing, leave as is */
    h264o_deblock_picture(d);
    mark_reference(d);
    h264o_pic *f = d->curf;
    if (d->field_pic) {
        /* a field: the frame goes out when its second field is done -- or, for a field that stays single, when the next picture starts */
        f->fields |= 1 << d->bottom;
        f->fpoc[d->bottom] = d->cur->poc;
        f->n_mbs = 0; /* (a B FRAME picture cannot take a field-coded frame as its co-located picture here) */
        if (f->ref) { /* a later B field may take this field as its co-located picture (RefPicList1[0]) */
            h264o_pic *v = d->cur;
            v->mbs = (h264o_mb *)realloc(v->mbs, sizeof(h264o_mb) * (size_t)d->wmb * d->fhmb);
            v->n_mbs = d->wmb * d->hmb;
            memcpy(v->mbs, d->mb, sizeof(h264o_mb) * (size_t)v->n_mbs);
        }
        d->cur = d->curf = NULL;
        if (f->fields == 3) {
            f->poc = f->fpoc[0] < f->fpoc[1] ? f->fpoc[0] : f->fpoc[1];
            emit_frame(d, f);
            f->in_use = 0;
            d->pend = NULL;
        } else
            d->pend = f;
        return;
    }
    /* the two fields of a frame picture have their own counts (8.2.1: bottom = top + delta_pic_order_cnt_bottom, or + offset_for_top_to_bottom_field);
     * operation 5 has made them relative to the frame's */
    const int mmco5 = d->first_sh.nal_ref_idc && !d->first_sh.idr_flag && f->poc == 0 && (d->poc_top || d->poc_bot);
    f->fields = 3, f->fpoc[0] = mmco5 ? d->poc_top - (d->poc_top < d->poc_bot ? d->poc_top : d->poc_bot) : d->poc_top, f->fpoc[1] = mmco5 ? d->poc_bot - (d->poc_top < d->poc_bot ? d->poc_top : d->poc_bot) : d->poc_bot;
    if (!d->asps->frame_mbs_only_flag) make_view(d, f, 0, f->fpoc[0]), make_view(d, f, 1, f->fpoc[1]); /* later field pictures may predict from its fields */
    if (f->ref) { /* a later B picture may use this one as its co-located picture (RefPicList1[0]) */
        if (f->n_mbs != d->wmb * d->hmb) {
            free(f->mbs);
            f->mbs = (h264o_mb *)malloc(sizeof(h264o_mb) * (size_t)d->wmb * d->hmb);
            f->n_mbs = d->wmb * d->hmb;
        }
        memcpy(f->mbs, d->mb, sizeof(h264o_mb) * (size_t)d->wmb * d->hmb);
    }
    emit_frame(d, f);
    f->in_use = 0;
    d->cur = d->curf = NULL;
}
/* a first field whose second field did not come: the frame goes out with one field decoded (the other half keeps the grey it was given) */
static void flush_pending_field(h264o_decoder *d) {
    h264o_pic *f = d->pend;
    if (!f) return;
    f->poc = f->fpoc[f->fields == 2];
    emit_frame(d, f);
    f->in_use = 0;
    d->pend = NULL;
}

/* 7.4.1.2.4 first VCL NAL of a new primary coded picture */
static int is_new_picture(const h264o_decoder *d, const h264o_slice_header *a, const h264o_slice_header *b) {
    if (a->frame_num != b->frame_num || a->pic_parameter_set_id != b->pic_parameter_set_id) return 1;
    if ((a->nal_ref_idc == 0) != (b->nal_ref_idc == 0)) return 1;
    if (a->field_pic_flag != b->field_pic_flag || a->bottom_field_flag != b->bottom_field_flag) return 1;
    if (a->idr_flag != b->idr_flag) return 1;
    if (a->idr_flag && a->idr_pic_id != b->idr_pic_id) return 1;
    if (d->asps->pic_order_cnt_type == 0 &&
        (a->pic_order_cnt_lsb != b->pic_order_cnt_lsb || a->delta_pic_order_cnt_bottom != b->delta_pic_order_cnt_bottom))
        return 1;
    if (d->asps->pic_order_cnt_type == 1 &&
        (a->delta_pic_order_cnt[0] != b->delta_pic_order_cnt[0] || a->delta_pic_order_cnt[1] != b->delta_pic_order_cnt[1]))
        return 1;
    /* beyond the list of 7.4.1.2.4: what 7.4.3 requires to be the same in all slice headers of a picture -- the slice group change
     * cycle and the marking script; they separate pictures whose frame_num and picture order count agree (after operation 5) */
    if (a->slice_group_change_cycle != b->slice_group_change_cycle) return 1;
    if (a->adaptive_ref_pic_marking_mode_flag != b->adaptive_ref_pic_marking_mode_flag || a->n_mmco != b->n_mmco) return 1;
    for (int k = 0; k < a->n_mmco; k++)
        if (a->mmco_op[k] != b->mmco_op[k] || a->mmco_arg1[k] != b->mmco_arg1[k] || a->mmco_arg2[k] != b->mmco_arg2[k]) return 1;
    return 0;
}

/* 8.2.5.2 decoding process for gaps in frame_num (h264/sps.go:311-312 parses the flag, nothing in the reference uses it):
 * frame_num of a non-IDR picture that is neither PrevRefFrameNum nor its successor means reference frames are missing.
 * With gaps_in_frame_num_value_allowed_flag every skipped value becomes a "non-existing" short-term frame that goes through
 * the sliding window like a decoded one (it pushes older frames out and takes its place in the initial lists, so that
 * the indices of the surviving pictures come out as the encoder meant them); without the flag pictures were lost. */
static int fill_frame_num_gap(h264o_decoder *d) {
    const h264o_sps *s = d->asps;
    const int max_fn = 1 << (s->log2_max_frame_num_minus4 + 4), cur_fn = d->sh.frame_num;
    if (d->sh.idr_flag) return 0;
    const int expect = (d->prev_ref_frame_num + 1) % max_fn;
    if (cur_fn == d->prev_ref_frame_num || cur_fn == expect) return 0;
    if (!s->gaps_in_frame_num_value_allowed_flag) return h264o_fail(d, "frame_num %d after %d: reference pictures are missing", cur_fn, d->prev_ref_frame_num);
    const int maxref = s->max_num_ref_frames > 0 ? s->max_num_ref_frames : 1;
    for (int fn = expect; fn != cur_fn; fn = (fn + 1) % max_fn) {
        int nref = 0;
        h264o_pic *oldest = NULL, *slot = NULL;
        for (int i = 0; i < d->n_pics; i++) { /* 8.2.5.3 with this frame as the current one */
            h264o_pic *p = &d->pics[i];
            if (!p->ref) continue;
            nref++;
            if (p->ref == 1) {
                p->frame_num_wrap = p->frame_num > fn ? p->frame_num - max_fn : p->frame_num;
                if (!oldest || p->frame_num_wrap < oldest->frame_num_wrap) oldest = p;
            }
        }
        if (nref >= maxref && oldest) oldest->ref = 0;
        for (int i = 0; i < d->n_pics && !slot; i++)
            if (!d->pics[i].ref && !d->pics[i].in_use) slot = &d->pics[i];
        if (!slot) return h264o_fail(d, "DPB full");
        slot->ref = 1, slot->nonexisting = 1, slot->frame_num = fn, slot->id = d->next_pic_id++, slot->n_mbs = 0, slot->funref = 0, slot->fields = 3;
        slot->poc = 0;
        if (s->pic_order_cnt_type != 0) { /* 8.2.1: as a reference frame with this frame_num (keeps FrameNumOffset right across a wrap) */
            h264o_slice_header f;
            memset(&f, 0, sizeof(f));
            f.frame_num = fn, f.nal_ref_idc = 1;
            slot->poc = compute_poc(d, &f);
        }
        d->prev_ref_frame_num = fn;
        d->feat |= 1u << 15;
    }
    return 0;
}

/* the rows of one parity of a frame store as a picture of their own (h264o_decoder::fviews) */
static h264o_pic *make_view(h264o_decoder *d, h264o_pic *p, int parity, int poc) {
    h264o_pic *v = &d->fviews[p - d->pics][parity];
    struct h264o_mb_s *keep = v->mbs; /* the motion array is kept for re-use; n_mbs = 0: nothing valid in it */
    memset(v, 0, sizeof(*v));
    v->mbs = keep;
    for (int i = 0; i < 3; i++) v->plane[i] = p->plane[i] + (size_t)parity * p->stride[i], v->stride[i] = 2 * p->stride[i];
    v->parity = parity, v->poc = poc, v->frame_num = p->frame_num, v->ref = 1;
    v->id = d->next_pic_id++;
    return v;
}

static int start_picture(h264o_decoder *d) {
    const h264o_slice_header *sh = &d->sh;
    d->field_pic = sh->field_pic_flag, d->bottom = sh->bottom_field_flag;
    d->hmb = d->field_pic ? d->fhmb / 2 : d->fhmb;
    d->scan4 = d->field_pic ? h264o_fieldscan4x4 : h264o_zigzag4x4;
    d->scan8 = d->field_pic ? h264o_fieldscan8x8 : h264o_zigzag8x8;
    /* the second field of the frame whose first field was the previous picture (7.4.1.2.4, 3.30 / 3.31): opposite parity, same frame_num,
     * not an IDR picture, reference or not like the first one */
    h264o_pic *p = NULL;
    d->second_field = 0;
    if (d->pend) {
        h264o_pic *f = d->pend;
        if (d->field_pic && !sh->idr_flag && f->fields == (d->bottom ? 1 : 2) && f->frame_num == sh->frame_num && (f->ref != 0) == (sh->nal_ref_idc != 0)) {
            p = f;
            d->second_field = 1;
            d->pend = NULL;
        } else
            flush_pending_field(d);
    }
    if (!p) {
        if (fill_frame_num_gap(d) < 0) return -1;
        for (int i = 0; i < d->n_pics; i++)
            if (!d->pics[i].ref && !d->pics[i].in_use) {
                p = &d->pics[i];
                break;
            }
        if (!p) return h264o_fail(d, "DPB full");
        p->in_use = 1;
        p->nonexisting = 0;
        p->fields = 0, p->funref = 0;
        p->id = d->next_pic_id++;
        p->frame_num = sh->frame_num;
        /* deterministic content for MBs that no slice covers (and for the field that may never come) */
        memset(p->plane[0], 128, (size_t)d->wmb * 16 * d->fhmb * 16 * 3 / 2);
    }
    d->curf = d->cur = p;
    const int poc = compute_poc(d, sh);
    if (d->field_pic) /* reconstruct through the view of this parity */
        d->cur = make_view(d, p, d->bottom, poc);
    else
        p->poc = poc;
    if (d->sh.slice_qp_delta) d->feat |= 1u << 13;
    if (d->asps->pic_order_cnt_type == 1 && d->sh.delta_pic_order_cnt[0]) d->feat |= 1u << 14;
    for (int i = 0; i < d->wmb * d->hmb; i++) d->mb[i].type = MBT_NONE;
    d->first_sh = d->sh;
    d->slice_id = 0;
    d->n_first_mbs = 0;
    /* slice groups: the macroblock-to-slice-group map of this picture (8.2.2; h264/slice.go:134-158) */
    free(d->sgmap);
    d->sgmap = NULL;
    if (d->apps->num_slice_groups_minus1 > 0) {
        d->sgmap = (uint8_t *)malloc((size_t)d->wmb * d->hmb);
        if (h264o_mb_to_slice_group_map(d->asps, d->apps, d->sg_ids[d->apps->pic_parameter_set_id], d->sh.slice_group_change_cycle, d->field_pic, d->sgmap) != d->wmb * d->hmb)
            return h264o_fail(d, "slice group map: bad PPS %d", d->apps->pic_parameter_set_id);
    }
    return 0;
}

static int decode_slice_nal(h264o_decoder *d, const h264o_nal *nal, const uint8_t *rbsp, size_t rlen) {
    h264o_br_init(&d->br, rbsp, rlen);
    h264o_slice_header sh;
    int r = h264o_parse_slice_header(&d->br, nal->nal_ref_idc, nal->nal_unit_type, d->sps, d->pps, &sh);
    if (r < 0) return h264o_fail(d, "slice header parse error %d", r);
    /* field pictures (PAFF): I, P and B fields, CAVLC and CABAC (the context values of field-coded blocks, ctxIdx 277..398 and 436..459, are UNPINNED:
     * h264o_cabac_mn.c), sliding-window marking and initial lists.  Out of scope for now: marking operations other than 1, and a
     * co-located picture of the other shape (a B field whose RefPicList1[0] belongs to a frame-coded frame, or the reverse) */
    if (sh.slice_type > 2) return h264o_fail(d, "slice_type %d out of scope (I, P and B only)", sh.slice_type);
    if (sh.redundant_pic_cnt > 0) return 0; /* redundant coded pictures are ignored */
    const h264o_pps *pps = &d->pps[sh.pic_parameter_set_id];
    /* (a slice that starts where a slice of the current picture already started begins a new picture whatever the headers say:
     * 7.4.1.2.4 cannot separate pictures whose headers agree, e.g. POC type 2 and equal frame_num after operation 5; with slice
     * groups or arbitrary slice order the slice of macroblock 0 may come late, so that one alone is no criterion) */
    int restarts = 0;
    for (int i = 0; i < d->n_first_mbs; i++) restarts |= d->first_mbs[i] == sh.first_mb_in_slice;
    if (d->cur && (restarts || is_new_picture(d, &d->first_sh, &sh))) finish_picture(d);
    if (!d->cur && d->pend && (d->asps != &d->sps[pps->seq_parameter_set_id])) flush_pending_field(d); /* (before the frame store may be rebuilt) */
    if (activate(d, pps) < 0) return -1;
    d->sh = sh;
    if (sh.slice_qp_delta) d->feat |= 1u << 13;
    if (!d->cur) {
        if (start_picture(d) < 0) return -1;
    } else
        d->slice_id++;
    if (sh.slice_type != 2 && build_ref_list(d) < 0) return -1;
    if (sh.first_mb_in_slice >= d->wmb * d->hmb) return h264o_fail(d, "first_mb_in_slice out of range");
    if (d->n_first_mbs < 1024) d->first_mbs[d->n_first_mbs++] = sh.first_mb_in_slice;
    return h264o_decode_slice_data(d) < 0 ? -1 : 0;
}

/* h264/server.go:113-166 handleConnection: NAL type dispatch (7 SPS, 8 PPS, 1/5 slices; the rest
 * ignored).  Parameter sets are looked up by id (the reference keeps only "the last": A33). */
int h264o_decode_stream(h264o_decoder *d, const uint8_t *buf, size_t len, int crop, uint8_t *out, size_t out_cap, h264o_stream_info *info) {
    int cap = 1 << 16, n;
    h264o_nal *nals = (h264o_nal *)malloc(sizeof(h264o_nal) * cap);
    n = h264o_annexb_scan(buf, len, nals, cap);
    memset(&d->info, 0, sizeof(d->info));
    d->err[0] = 0;
    d->crop = crop;
    d->out = out;
    d->out_cap = out_cap;
    d->out_pos = 0;
    d->trace_pos = 0;
    d->feat = 0, d->n_pocs = 0;
    for (int i = 0; i < d->n_pics; i++) d->pics[i].ref = 0, d->pics[i].in_use = 0, d->pics[i].fields = 0;
    d->cur = d->curf = d->pend = NULL;
    for (int i = 0; i < n && !d->info.error; i++) {
        const h264o_nal *nal = &nals[i];
        if (nal->size > d->rbsp_cap) {
            d->rbsp_cap = nal->size * 2;
            d->rbsp = (uint8_t *)realloc(d->rbsp, d->rbsp_cap);
        }
        size_t rlen = h264o_nal_to_rbsp(buf + nal->offset, nal->size, d->rbsp);
        switch (nal->nal_unit_type) {
        case 7: {
            h264o_sps s;
            if (h264o_parse_sps(d->rbsp, rlen, &s) == 0) {
                if (d->cur) finish_picture(d);
                if (memcmp(&d->sps[s.seq_parameter_set_id], &s, sizeof(s)) != 0) { /* repeated identical SPS: keep the DPB */
                    if (d->asps == &d->sps[s.seq_parameter_set_id]) d->asps = NULL; /* force re-activation */
                    d->sps[s.seq_parameter_set_id] = s;
                }
            } else
                h264o_fail(d, "SPS parse error");
            break;
        }
        case 8: {
            h264o_pps p;
            size_t n_ids = 0, cap = 1 << 20;
            uint8_t *ids = (uint8_t *)malloc(cap);
            int r = h264o_parse_pps_ids(d->rbsp, rlen, d->sps, &p, ids, cap, &n_ids);
            if (r == 0 && n_ids <= cap) {
                if (d->cur) finish_picture(d);
                d->pps[p.pic_parameter_set_id] = p;
                free(d->sg_ids[p.pic_parameter_set_id]);
                d->sg_ids[p.pic_parameter_set_id] = n_ids ? ids : NULL;
                if (!n_ids) free(ids);
            } else if (r == 0) {
                free(ids);
                h264o_fail(d, "PPS: slice group map too large");
            } else {
                free(ids);
                h264o_fail(d, "PPS parse error %d", r);
            }
            break;
        }
        case 1:
        case 5: decode_slice_nal(d, nal, d->rbsp, rlen); break;
        case 9: /* access unit delimiter */
        case 10:
        case 11:
            if (d->cur) finish_picture(d);
            break;
        default: break; /* SEI, filler, ... ignored (h264/server.go:147-164) */
        }
    }
    if (d->cur && !d->info.error) finish_picture(d);
    if (!d->info.error) flush_pending_field(d);
    free(nals);
    if (info) *info = d->info;
    if (d->info.error) return -1;
    if (out && d->out_pos > out_cap) return -2;
    return 0;
}

uint32_t h264o_last_features(h264o_decoder *d) { return d->feat; }
int h264o_last_pocs(h264o_decoder *d, int32_t *dst, int cap) {
    for (int i = 0; i < d->n_pocs && i < cap && dst; i++) dst[i] = d->pocs[i];
    return d->n_pocs;
}

// h264decode_amd/csrc/k_dbprep.hip -- k_dbprep: boundary strengths (8.7.2.1) and alpha / beta / tC0 (8.7.2.2) of every macroblock of a batch, gfx950.
//
// Neither depends on samples, so they are worked out here for all macroblocks at once, fully parallel, and left as an 80-byte DbPrm per
// macroblock; the deblocking kernels (k_deblock.hip, k_deblock_x.hip) -- a serial dependency chain per picture -- only pick their bytes out of
// it.  (Pictures with B slices differ in the strengths only, so they need no deblocking kernel of their own.)  The same pass leaves K3's work
// list (one bit per intra macroblock) and the ColRec arrays later B pictures take their direct prediction from.
// Absent from the reference (only the slice-header fields are parsed: h264/slice.go:1021-1027).
#include <hip/hip_runtime.h>
#include "mi_kernels.h"

#define WAVE_SYNC()                                            \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

// 8.7.2.1 with one list (I / P pictures)
// (strong: the bS of an intra macroblock edge -- 4, but 3 on the horizontal macroblock edges of a field picture; vlim: the vertical vector
// difference that counts as "far" -- 4 quarter frame samples = 2 quarter field samples in a field picture)
__device__ __forceinline__ int prep_bs(const MbRec *mp, int pb, const MbRec *mq, int qb, int strong, int vlim) {
    // branch-free: every operand is fetched up front (independent LDS reads), the decision is a chain of selects
    const int q8p = ((pb >> 3) << 1) | ((pb & 3) >> 1), q8q = ((qb >> 3) << 1) | ((qb & 3) >> 1);
    const int tp = mp->type, tq = mq->type, nzp = mp->nzmask, nzq = mq->nzmask, rp = mp->refslot[q8p], rq = mq->refslot[q8q];
    const int vpx = mp->mv[pb][0], vpy = mp->mv[pb][1], vqx = mq->mv[qb][0], vqy = mq->mv[qb][1];
    const bool far = rp != rq || abs(vpx - vqx) >= 4 || abs(vpy - vqy) >= vlim;
    return (MB_IS_INTRA(tp) || MB_IS_INTRA(tq)) ? strong : ((((nzp >> pb) | (nzq >> qb)) & 1) ? 2 : (far ? 1 : 0));
}
// 8.7.2.1 with two lists (pictures with B slices): the blocks differ if they use different reference PICTURES (frame slots; the
// list a picture comes from does not matter) or a different number of vectors, or if the vectors that belong together differ by >= 4
__device__ __forceinline__ bool mv_far(const int16_t *a, const int16_t *b, int vlim) { return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= vlim; }
__device__ __forceinline__ int prep_bs_b(const MbRec *mp, const MbMv1 *vp, int pb, const MbRec *mq, const MbMv1 *vq, int qb, int strong, int vlim) {
    if (MB_IS_INTRA(mp->type) || MB_IS_INTRA(mq->type)) return strong;
    if (((mp->nzmask >> pb) & 1) || ((mq->nzmask >> qb) & 1)) return 2;
    const int p8 = ((pb >> 3) << 1) | ((pb & 3) >> 1), q8 = ((qb >> 3) << 1) | ((qb & 3) >> 1);
    const int p0 = mp->refslot[p8], p1 = mp->refslot1[p8], q0 = mq->refslot[q8], q1 = mq->refslot1[q8];
    const int np = (p0 >= 0) + (p1 >= 0), nq = (q0 >= 0) + (q1 >= 0);
    if (np != nq) return 1;
    const int16_t *pv0 = mp->mv[pb], *pv1 = vp->mv[pb], *qv0 = mq->mv[qb], *qv1 = vq->mv[qb];
    if (np < 2) { // one vector each (or none: corrupt records)
        const int rp = p0 >= 0 ? p0 : p1, rq = q0 >= 0 ? q0 : q1;
        if (rp != rq) return 1;
        return mv_far(p0 >= 0 ? pv0 : pv1, q0 >= 0 ? qv0 : qv1, vlim) ? 1 : 0;
    }
    if (!((p0 == q0 && p1 == q1) || (p0 == q1 && p1 == q0))) return 1;
    if (p0 != p1) // two different pictures: each vector against the one that points to the same picture
        return (p0 == q0 ? (mv_far(pv0, qv0, vlim) || mv_far(pv1, qv1, vlim)) : (mv_far(pv0, qv1, vlim) || mv_far(pv1, qv0, vlim))) ? 1 : 0;
    return ((mv_far(pv0, qv0, vlim) || mv_far(pv1, qv1, vlim)) && (mv_far(pv0, qv1, vlim) || mv_far(pv1, qv0, vlim))) ? 1 : 0; // both vectors into one picture
}

struct PrepSub {
    MbRec rec[3]; // current, left, upper macroblock
    MbMv1 mv1[3]; // their list-1 vectors (pictures with B slices)
    DbPrm out;
    ColRec col;   // what later B pictures need of this macroblock's motion (pictures flagged save_col)
};
// grid = (ceil(macroblocks of the largest picture / MI_DBPREP_MBS), pictures), block = 256: a wavefront works on 4 macroblocks at a
// time, 16 lanes each -- lane li computes the strength of segment li & 3 of vertical edge li >> 2 and of horizontal edge li >> 2
// (the same division of labour K5 had when it did this itself), lanes 0..8 the parameters of (plane, edge kind) li / 3, li % 3.
// The same pass leaves the ColRec array of the pictures a later B picture (or batch) may take as co-located picture
// (8.4.1.2.1: per 4x4 block the vector of the list the block uses -- list 0 if it uses it, otherwise list 1 --, per 8x8 the
// reference index and the frame slot of the picture it points to; -1: intra): the records are staged here anyway.
// (Measured in round 4: issuing the loads of the wavefront's next step before working on the current one -- 7.8 -> 9.3 ms per 7680 pictures; twice the
// macroblocks per workgroup on top of that -- 8.9 ms.  A million short workgroups hide the dependent loads better than a loop carrying 12 registers.)
// col_only: the one-off back-fill of ColRec arrays for a batch whose DbPrm records are already in use (mi_api.cpp: ensure_b_buffers).
extern "C" __global__ void __launch_bounds__(256) k_dbprep(const uint32_t *pic_list, const PicDesc *pics, const DevTables *tab, const MbRec *mbrec, const MbMv1 *mbmv1,
                                                           DbPrm *out, int col_only, unsigned long long *intramask, int n_pics) {
    __shared__ PrepSub subs[4][4];
    __shared__ uint8_t s_alpha[52], s_beta[52], s_tc0[52][4];
    const int tid = static_cast<int>(threadIdx.x), wave = tid >> 6, lane = tid & 63, sub = lane >> 4, li = lane & 15;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, each with its own L2, and a macroblock's upper neighbour is a record another
    // workgroup of the picture loads as its own -- so an XCD takes WHOLE pictures (picture p goes to XCD p mod 8; grid.y is a multiple of 8), and the row
    // above comes out of the L2 that fetched it a moment ago
    const uint32_t w = blockIdx.y * gridDim.x + blockIdx.x, in_xcd = w >> 3;
    const uint32_t pic = (in_xcd / gridDim.x) * 8u + (w & 7u), blk = in_xcd % gridDim.x;
    if (pic >= static_cast<uint32_t>(n_pics)) return;
    const PicDesc *pd = &pics[pic_list[pic]];
    const int wmb = static_cast<int>(pd->wmb), nmb = wmb * static_cast<int>(pd->hmb);
    const int mb_first = static_cast<int>(blk) * MI_DBPREP_MBS;
    if (mb_first >= nmb) return;
    for (int i = tid; i < 52; i += 256) {
        s_alpha[i] = tab->alpha[i], s_beta[i] = tab->beta[i];
        s_tc0[i][0] = 0, s_tc0[i][1] = tab->tc0[i][1], s_tc0[i][2] = tab->tc0[i][2], s_tc0[i][3] = tab->tc0[i][3];
    }
    __syncthreads();
    const bool two = pd->has_b != 0;
    const MbRec *recs = mbrec + pd->mb_base;
    const MbMv1 *recs1 = two ? mbmv1 + pd->mb_base : nullptr;
    DbPrm *outs = out + pd->mb_base;
    PrepSub *ss = &subs[wave][sub];
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    for (int it = wave; it < MI_DBPREP_MBS / 4; it += 4) {
        const int mb = mb_first + it * 4 + sub;
        const bool valid = mb < nmb;
        const int mby = static_cast<int>(__umulhi(static_cast<uint32_t>(valid ? mb : 0), pd->inv_wmb)), mbx = (valid ? mb : 0) - mby * wmb;
        const bool has_left = valid && mbx > 0, has_top = valid && mby > 0;
        if (valid) { // lanes 0-7: the record, lanes 8-15: the record above; then lanes 0-7: the record to the left
            const v4u z = v4u{0u, 0u, 0u, 0u};
            const bool up = li >= 8;
            v4u a = z, b = z;
            if (!up || has_top) a = reinterpret_cast<const v4u *>(recs + (up ? mb - wmb : mb))[li & 7];
            if (!up && has_left) b = reinterpret_cast<const v4u *>(recs + mb - 1)[li];
            reinterpret_cast<v4u *>(&ss->rec[up ? 2 : 0])[li & 7] = a;
            if (!up) reinterpret_cast<v4u *>(&ss->rec[1])[li] = b;
            if (two) { // list-1 vectors: lanes 0-3 current, 4-7 left, 8-11 above
                const int which = li >> 2;
                if (which < 3) {
                    const bool ok = which == 0 || (which == 1 ? has_left : has_top);
                    reinterpret_cast<v4u *>(&ss->mv1[which == 0 ? 0 : (which == 1 ? 1 : 2)])[li & 3] =
                        ok ? reinterpret_cast<const v4u *>(recs1 + (which == 0 ? mb : (which == 1 ? mb - 1 : mb - wmb)))[li & 3] : z;
                }
            }
        }
        WAVE_SYNC();
        // K3's work list: one bit per macroblock of the batch (index = position in the record array), set for intra macroblocks and for
        // macroblocks no slice delivered -- K3 reads 1 KB per 1080p picture instead of one type byte out of every 128-byte record
        if (valid && !col_only && li == 0 && (MB_IS_INTRA(ss->rec[0].type) || ss->rec[0].type == MBT_NONE)) {
            const unsigned long long gmb = pd->mb_base + static_cast<unsigned long long>(mb);
            atomicOr(&intramask[gmb >> 6], 1ull << (gmb & 63));
        }
        if (valid) {
            const MbRec *mq = &ss->rec[0], *ml = has_left ? &ss->rec[1] : nullptr, *mt = has_top ? &ss->rec[2] : nullptr;
            const int dbf = mq->dbf_idc;
            if (dbf == 2) { // no filtering across slice boundaries
                if (ml && ml->slice_in_pic != mq->slice_in_pic) ml = nullptr;
                if (mt && mt->slice_in_pic != mq->slice_in_pic) mt = nullptr;
            }
            const int e = li >> 2, k = li & 3;
            const bool mb_edge = e == 0;
            const int qb0 = k * 4 + e, qb1 = li; // q block of the vertical / horizontal edge segment
            const int pb0 = mb_edge ? k * 4 + 3 : qb0 - 1, pb1 = mb_edge ? 12 + k : qb1 - 4;
            const bool ok = dbf != 1 && !((e & 1) && mq->t8x8);
            const bool ok0 = ok && !(mb_edge && !ml), ok1 = ok && !(mb_edge && !mt);
            const MbRec *mp0 = mb_edge && ml ? ml : mq, *mp1 = mb_edge && mt ? mt : mq; // (no neighbour: any record, the result is masked)
            int bs0, bs1;
            // 8.7.2.1 in a field picture: bS 4 needs a VERTICAL macroblock edge (horizontal ones get 3), and vectors differ from a
            // vertical distance of 4 quarter FRAME samples on = 2 quarter field samples
            const bool fieldpic = pd->field != 0;
            const int vlim = fieldpic ? 2 : 4, strong0 = mb_edge ? 4 : 3, strong1 = mb_edge && !fieldpic ? 4 : 3;
            if (two) {
                bs0 = prep_bs_b(mp0, mb_edge && ml ? &ss->mv1[1] : &ss->mv1[0], pb0, mq, &ss->mv1[0], qb0, strong0, vlim);
                bs1 = prep_bs_b(mp1, mb_edge && mt ? &ss->mv1[2] : &ss->mv1[0], pb1, mq, &ss->mv1[0], qb1, strong1, vlim);
            } else
                bs0 = prep_bs(mp0, pb0, mq, qb0, strong0, vlim), bs1 = prep_bs(mp1, pb1, mq, qb1, strong1, vlim);
            ss->out.bs[k][0][e] = static_cast<uint8_t>(ok0 ? bs0 : 0);
            ss->out.bs[k][1][e] = static_cast<uint8_t>(ok1 ? bs1 : 0);
            if (li < 9) { // 8.7.2.2: (plane, edge kind): qPav of the left / no / the upper neighbour, indexA / indexB, the table rows
                const int plane = li / 3, kind = li - plane * 3;
                const MbRec *mn = kind == 0 ? ml : (kind == 2 ? mt : nullptr);
                const int qpq = plane == 0 ? mq->qp : mq->qpc[plane - 1];
                const int qpn = mn ? (plane == 0 ? mn->qp : mn->qpc[plane - 1]) : qpq;
                const int qpav = (qpn + qpq + 1) >> 1;
                const int ia = min(max(qpav + mq->alpha_off, 0), 51), ib = min(max(qpav + mq->beta_off, 0), 51);
                ss->out.pl[plane].ab[2 * kind] = s_alpha[ia], ss->out.pl[plane].ab[2 * kind + 1] = s_beta[ib];
                ss->out.pl[plane].tc[kind][0] = s_tc0[ia][1], ss->out.pl[plane].tc[kind][1] = s_tc0[ia][2], ss->out.pl[plane].tc[kind][2] = s_tc0[ia][3];
                if (kind == 0) ss->out.pl[plane].pad = 0;
            }
            if (pd->save_col) { // lane li: block li's vector; lanes 0..3 also the reference of 8x8 quadrant li
                const bool inter = MB_IS_INTER(mq->type);
                const int q = ((li >> 3) << 1) | ((li & 3) >> 1);
                const bool l0 = inter && mq->ref[q] >= 0, l1 = inter && !l0 && two && mq->refslot1[q] >= 0;
                ss->col.mv[li][0] = l0 ? mq->mv[li][0] : (l1 ? ss->mv1[0].mv[li][0] : static_cast<int16_t>(0));
                ss->col.mv[li][1] = l0 ? mq->mv[li][1] : (l1 ? ss->mv1[0].mv[li][1] : static_cast<int16_t>(0));
                if (li < 4) {
                    const bool q0 = inter && mq->ref[li] >= 0, q1 = inter && !q0 && two && mq->refslot1[li] >= 0;
                    ss->col.refslot[li] = q0 ? mq->refslot[li] : (q1 ? mq->refslot1[li] : static_cast<int16_t>(-1));
                    ss->col.ref[li] = q0 ? mq->ref[li] : (q1 ? MBREC_REF1(mq)[li] : static_cast<int8_t>(-1));
                    ss->col.pad[li] = 0;
                }
            }
        }
        WAVE_SYNC();
        if (valid && !col_only && li < 5) reinterpret_cast<v4u *>(outs + mb)[li] = reinterpret_cast<const v4u *>(&ss->out)[li];
        if (valid && pd->save_col && li < 5) reinterpret_cast<v4u *>(reinterpret_cast<ColRec *>(pd->col_out) + mb)[li] = reinterpret_cast<const v4u *>(&ss->col)[li];
        WAVE_SYNC();
    }
}

"""GPU bring-up / debugging helper (not a test): decodes synthetic streams on the GPU and compares
against the stream generator's reconstruction and the oracle, printing the first divergence in
detail (macroblock record fields first, then samples)."""
import sys
import os
import time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import streamgen  # noqa: E402
import oracle  # noqa: E402
import h264decode_amd as H  # noqa: E402

MBT = ["NONE", "I4x4", "I8x8", "I16x16", "IPCM", "P16x16", "P16x8", "P8x16", "P8x8", "PSKIP", "B", "BDIRECT", "BSKIP"]


def _nslices(kw):
    """slices per picture of a parity-matrix case (with slice groups: `slices` per group)"""
    return max(1, kw.get("slices", 1)) * max(1, kw.get("slice_groups", 1))


def gpu_type_from_trace(raw, islice_guess):
    return None


def compare(name, kw, dec_cache={}):
    s, rec, sizes = streamgen.encode(**kw)
    W, H_ = (kw["width"] + 15) // 16 * 16, (kw["height"] + 15) // 16 * 16
    nmb = (W // 16) * (H_ // 16)
    key = (W, H_)
    t0 = time.time()
    dec = H.Decoder(max_streams=1, max_width=W, max_height=H_, max_frames_per_batch=kw["frames"], max_slices_per_frame=max(1, _nslices(kw)))
    try:
        dec.decode([s])
    except Exception as ex:
        print(f"[{name}] GPU DECODE ERROR: {ex}")
        # still compare what we can
    n = dec.frame_count(0)
    out = dec.read_frames(0, crop=False, size=W * H_ * 3 // 2) if n else np.zeros((0, W * H_ * 3 // 2), np.uint8)
    ok = out.shape == rec.shape and np.array_equal(out, rec)
    print(f"[{name}] frames={n}/{kw['frames']} bytes={len(s)} {'OK' if ok else 'MISMATCH'} ({time.time()-t0:.1f}s)")
    if ok:
        dec.close()
        return True
    # --- detailed diagnosis ---
    oout, info, tr = oracle.decode(s, crop=False, trace=True)
    print("   oracle==streamgen:", np.array_equal(oout, rec))
    tr = tr.reshape(-1, nmb, 8)
    shown = 0
    for f in range(min(n, kw["frames"])):
        recs = dec.read_mbrecs(0, f, nmb)
        typ = recs[:, 0]
        t8 = recs[:, 1]
        qp = recs[:, 2]
        cbp = recs[:, 5]
        mv0 = recs[:, 48:52].copy().view(np.int16).reshape(-1, 2)
        ref0 = recs[:, 32].view(np.int8)
        for m in range(nmb):
            o = tr[f, m]
            raw = o[0]
            # expected GPU type from the oracle's raw mb_type
            is_i = (typ[m] >= 1 and typ[m] <= 4)
            bad = []
            if o[1] != cbp[m]:
                bad.append(f"cbp {o[1]:#x}!={cbp[m]:#x}")
            if o[2] != qp[m]:
                bad.append(f"qp {o[2]}!={qp[m]}")
            if o[4] != t8[m]:
                bad.append(f"t8x8 {o[4]}!={t8[m]}")
            if not is_i and ref0[m] >= 0 and (o[5] != mv0[m, 0] or o[6] != mv0[m, 1]):
                bad.append(f"mv0 ({o[5]},{o[6]})!=({mv0[m,0]},{mv0[m,1]})")
            oref = o[7] + 100 if o[7] <= -100 else o[7]  # the oracle marks macroblocks of B slices by ref - 100
            if not is_i and oref != ref0[m]:
                bad.append(f"ref0 {oref}!={ref0[m]}")
            if raw == -1 and typ[m] not in (9, 12):
                bad.append(f"type skip!={MBT[typ[m]] if typ[m] < 13 else typ[m]}")
            if bad and shown < 6:
                print(f"   frame {f} mb {m} ({m % (W//16)},{m // (W//16)}) raw={raw} gpu_type={MBT[typ[m]] if typ[m] < 13 else typ[m]}: " + "; ".join(bad))
                shown += 1
        if f < out.shape[0] and not np.array_equal(out[f], rec[f]):
            d = np.nonzero(out[f] != rec[f])[0]
            i = d[0]
            if i < W * H_:
                x, y = i % W, i // W
                m = (y // 16) * (W // 16) + x // 16
                pl = "Y"
            else:
                j = (i - W * H_) % (W * H_ // 4)
                x, y = j % (W // 2), j // (W // 2)
                m = (y // 8) * (W // 16) + x // 8
                pl = "Cb" if i < W * H_ * 5 // 4 else "Cr"
            ymis = int((out[f][:W * H_] != rec[f][:W * H_]).sum())
            cmis = len(d) - ymis
            print(f"   frame {f}: first sample mismatch {pl}({x},{y}) mb {m} type {MBT[typ[m]] if typ[m] < 13 else typ[m]} gpu={out[f][i]} ref={rec[f][i]}; "
                  f"mismatching luma={ymis} chroma={cmis}; mb types in frame: { {MBT[t]: int((typ==t).sum()) for t in np.unique(typ) if t < 13} }")
            # which MBs mismatch (luma)
            dy = (out[f][:W * H_] != rec[f][:W * H_]).reshape(H_ // 16, 16, W // 16, 16).any(axis=(1, 3))
            bad_mbs = np.nonzero(dy.reshape(-1))[0]
            print(f"   luma-mismatching MBs ({len(bad_mbs)}): {[(int(b), MBT[typ[b]] if typ[b] < 13 else int(typ[b])) for b in bad_mbs[:12]]}")
            if os.environ.get("DEBUG_MB"):  # sample-level picture of the first few macroblocks that differ (luma, then Cb)
                Yg, Yr = out[f][:W * H_].reshape(H_, W), rec[f][:W * H_].reshape(H_, W)
                Cg, Cr_ = out[f][W * H_:W * H_ * 5 // 4].reshape(H_ // 2, W // 2), rec[f][W * H_:W * H_ * 5 // 4].reshape(H_ // 2, W // 2)
                dc = (Cg != Cr_).reshape(H_ // 16, 8, W // 16, 8).any(axis=(1, 3)).reshape(-1)
                for bmb in list(bad_mbs[:2]) + [int(x) for x in np.nonzero(dc)[0][:1]]:
                    mx, my = bmb % (W // 16), bmb // (W // 16)
                    r_ = recs[bmb]
                    print(f"   mb {bmb} ({mx},{my}) type {typ[bmb]} t8x8 {t8[bmb]} cbp {cbp[bmb]:#x} qp {qp[bmb]} mv0 {r_[48:112].copy().view(np.int16).reshape(16, 2)[:4].tolist()}.. refslot {r_[36:44].copy().view(np.int16).tolist()} cmask {int(r_[116:120].copy().view(np.uint32)[0]):#x}")
                    g_, w_ = Yg[my * 16:my * 16 + 16, mx * 16:mx * 16 + 16].astype(int), Yr[my * 16:my * 16 + 16, mx * 16:mx * 16 + 16].astype(int)
                    print("   luma got - want:")
                    for rr in range(16):
                        print("     " + " ".join("%4d" % v for v in (g_ - w_)[rr]))
                    g_, w_ = Cg[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8].astype(int), Cr_[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8].astype(int)
                    print("   Cb got - want:")
                    for rr in range(8):
                        print("     " + " ".join("%4d" % v for v in (g_ - w_)[rr]))
            break
    dec.close()
    return False


def main():
    base = dict(width=64, height=48, frames=3, idr_period=0)
    cases = [
        ("cavlc_I", dict(width=64, height=48, frames=1, idr_period=1, profile_idc=66, cabac=0)),
        ("cabac_I", dict(width=64, height=48, frames=1, idr_period=1, profile_idc=77, cabac=1)),
        ("cavlc_IPP", dict(**base, profile_idc=66, cabac=0)),
        ("cabac_IPP", dict(**base, profile_idc=77, cabac=1)),
        ("cabac_nodbf", dict(**base, profile_idc=77, cabac=1, deblock_idc=1)),
        ("cavlc_nodbf", dict(**base, profile_idc=66, cabac=0, deblock_idc=1)),
        ("high8x8_cabac", dict(**base, profile_idc=100, cabac=1, transform8x8=1)),
        ("high8x8_cavlc", dict(**base, profile_idc=100, cabac=0, transform8x8=1)),
        ("qcif_cabac", dict(width=176, height=144, frames=5, idr_period=0, profile_idc=77, cabac=1, qp=24)),
        ("qcif_cavlc", dict(width=176, height=144, frames=5, idr_period=0, profile_idc=66, cabac=0, qp=24)),
        ("slices_pcm_jit", dict(width=176, height=144, frames=4, idr_period=0, profile_idc=77, cabac=1, slices=3, pcm_permille=40, qp_jitter=5, cabac_init_idc=-1)),
        ("multiref_wp", dict(width=176, height=144, frames=5, idr_period=0, profile_idc=77, cabac=1, num_ref_frames=3, weighted_pred=1)),
        ("cip_sub8", dict(width=176, height=144, frames=4, idr_period=0, profile_idc=77, cabac=1, constrained_intra=1, intra_in_p_permille=250, sub8x8_permille=500)),
        ("scaling", dict(width=176, height=144, frames=3, idr_period=0, profile_idc=100, cabac=1, transform8x8=1, scaling_matrix=1)),
        ("lowqp", dict(width=176, height=144, frames=3, idr_period=0, profile_idc=77, cabac=1, qp=8, noise=30)),
        ("lowqp_cavlc", dict(width=176, height=144, frames=3, idr_period=0, profile_idc=66, cabac=0, qp=8, noise=30)),
        ("crop_dbf2", dict(width=180, height=100, frames=3, idr_period=0, profile_idc=77, cabac=1, slices=2, deblock_idc=2, alpha_off_div2=2, beta_off_div2=-1, chroma_qp_offset=3)),
    ]
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
    from conftest import MATRIX
    cases += [(k, v) for k, v in sorted(MATRIX.items()) if v.get("bframes")]
    sel = sys.argv[1:]
    nok = 0
    for name, kw in cases:
        if sel and name not in sel:
            continue
        nok += compare(name, kw)
    print(f"{nok} cases OK")


if __name__ == "__main__":
    main()

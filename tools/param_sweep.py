#!/usr/bin/env python3
"""Randomised sweep over the generator's feature space (not a test: a bug hunt).  Every trial draws a stream recipe -- profile, entropy
coder, slices, slice groups, references, picture management, B pictures, weights, motion, sizes -- and compares decoder output with the
generator's reconstruction bit for bit.  Usage: param_sweep.py [trials] [--gpu] [--seed N] [--batch B]
  without --gpu: the oracle (CPU);  with --gpu: the product through the C ABI, one workgroup per picture and banded;
  --batch B (GPU): B streams of different recipes and sizes side by side in one decoder per trial;
  --fields: field-picture (PAFF) recipes -- every frame as two field pictures, or picture-adaptively a frame or two fields; with --split the
  pieces are whole PICTURES, so the two fields of a frame may arrive in different calls;  --pocdelta: frame pictures whose bottom field has its own picture order count;
  --split (GPU): every stream is fed in several calls, a random number of access units at a time (state that must survive a batch boundary:
  reference pictures and their marking, picture order counts, co-located motion, frame_num gap bookkeeping, parameter sets)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import streamgen

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if args else 100
GPU = "--gpu" in sys.argv
seed0 = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 1
BATCH = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 1
SPLIT = "--split" in sys.argv
XR = "--xwgs" in sys.argv  # a random workgroup budget per trial (band plans of every shape) instead of the two standard ones
EXTREME = "--extreme" in sys.argv  # the corners of the value ranges: QP 0..51, chroma offsets -12..12, filter offsets -6..6, loud noise (escape-coded levels)
BIG = "--big" in sys.argv  # pictures wider than 64 macroblocks (rows of more than one 64-macroblock chunk), more slices
FIELDS = "--fields" in sys.argv  # PAFF recipes: every frame as two field pictures, or a frame / two fields picture by picture
POCD = "--pocdelta" in sys.argv  # bottom_field_pic_order_in_frame_present_flag = 1: the bottom field of every frame picture at its own count (before or after the top field)
CONCAT = "--concat" in sys.argv  # two recipes back to back in one stream: new parameter sets, entropy coder, slice groups, picture size at the second IDR picture
args = [a for a in args if a not in (str(seed0), str(BATCH))] or args[:1]
rng = np.random.default_rng(seed0)


def draw():
    r = lambda lo, hi: int(rng.integers(lo, hi + 1))
    pick = lambda *xs: xs[int(rng.integers(0, len(xs)))]
    prof = pick(66, 77, 77, 100, 100)
    kw = dict(width=16 * r(2, 13) - pick(0, 0, 4, 10), height=16 * r(2, 10) - pick(0, 0, 2, 6), frames=r(2, 9), profile_idc=prof, seed=r(1, 1 << 20),
              qp=r(10, 44), qp_jitter=pick(0, 0, 2, 5), idr_period=pick(0, 0, 1, 3, 5), slices=pick(1, 1, 2, 3), num_ref_frames=r(1, 4),
              deblock_idc=pick(0, 0, 1, 2), alpha_off_div2=r(-3, 3), beta_off_div2=r(-3, 3), constrained_intra=pick(0, 0, 1), chroma_qp_offset=r(-4, 4),
              pcm_permille=pick(0, 0, 20), intra_in_p_permille=pick(20, 50, 200), skip_permille=pick(100, 250, 500), sub8x8_permille=pick(50, 100, 400),
              noise=pick(2, 8, 20), long_start_code=pick(0, 1), motion_x4=r(-20, 20), motion_y4=r(-20, 20), slice_qp_delta=pick(0, 0, 3))
    kw["cabac"] = 0 if prof == 66 else pick(0, 1, 1)
    if kw["cabac"]:
        kw["cabac_init_idc"] = pick(-1, 0, 1, 2)
    if prof == 100:
        kw["transform8x8"], kw["scaling_matrix"] = pick(0, 1, 1), pick(0, 1)
        kw["mono"] = pick(0, 0, 0, 1)  # chroma_format_idc 0 now and then
    if prof != 66:
        kw["weighted_pred"] = pick(0, 0, 1, 2)
        if rng.random() < 0.4 and kw["frames"] >= 4:
            kw.update(bframes=r(1, 3), direct_temporal=pick(0, 1), weighted_bipred=pick(0, 1, 2), bskip_permille=pick(100, 300), b_pyramid=pick(0, 1))
            kw["num_ref_frames"] = max(kw["num_ref_frames"], 2)
    if not kw.get("bframes"):
        kw["poc_type"] = pick(0, 0, 1, 2)
        kw["rplm"], kw["mmco"], kw["idr_long_term"] = pick(0, 0, 1), pick(0, 0, 1), pick(0, 0, 1)
        kw["nonref_period"] = pick(0, 0, 3)
        if not (kw["rplm"] or kw["mmco"] or kw["idr_long_term"] or kw["nonref_period"]) and rng.random() < 0.3:
            kw.update(fn_gap_period=r(2, 4), fn_gap_declared=1)
    if rng.random() < 0.3:
        kw.update(slice_groups=r(2, 6), fmo_type=r(0, 6), aso=pick(0, 1))
        kw["slices"] = min(kw["slices"], 2)
    elif kw["slices"] > 1:
        kw["aso"] = pick(0, 1)
    if EXTREME:
        kw.update(qp=pick(0, 1, 2, 5, 48, 50, 51, r(0, 51)), chroma_qp_offset=pick(-12, -9, 9, 12, r(-12, 12)), alpha_off_div2=pick(-6, 6, r(-6, 6)), beta_off_div2=pick(-6, 6, r(-6, 6)),
                  noise=pick(0, 40, 60, 100), qp_jitter=pick(0, 5, 8, 12), slice_qp_delta=pick(0, 3, 8))
    if BIG:
        kw["width"], kw["height"], kw["frames"] = 16 * r(62, 84) - pick(0, 6), 16 * r(3, 9) - pick(0, 2), r(2, 4)
        kw["slices"] = pick(1, 2, 3, min(5, (kw["height"] + 15) // 16)) if not kw.get("slice_groups") else kw["slices"]
    if rng.random() < 0.15 or FIELDS:
        kw["interlace_sps"] = 1
        kw["height"] = max(32, (kw["height"] + 31) // 32 * 32 - pick(0, 4, 8))
    if FIELDS:  # what sg.h says field recipes may carry: Main / High, no long-term pictures; B fields in all-field streams; CABAC with the unpinned field contexts
        for k in ("b_pyramid", "idr_long_term", "fn_gap_period", "fn_gap_declared"):
            kw.pop(k, None)
        kw.update(field_pics=pick(1, 2, 3, 3), cabac=pick(0, 1), profile_idc=pick(77, 100))
        if not kw["cabac"]:
            kw.pop("cabac_init_idc", None)
        if kw["field_pics"] == 3 or not kw.get("bframes"):
            for k in ("bframes", "weighted_bipred", "bskip_permille", "direct_temporal"):
                kw.pop(k, None)
            kw["poc_type"] = pick(0, 0, 1, 2)
        if kw["profile_idc"] != 100:
            kw.pop("transform8x8", None), kw.pop("scaling_matrix", None), kw.pop("mono", None)
        else:
            kw["transform8x8"], kw["scaling_matrix"] = pick(0, 1, 1), pick(0, 1)
        kw["weighted_pred"] = pick(0, 0, 1, 2)
    if POCD:
        kw["poc_bottom_delta"] = pick(-1, -1, 1, 2, 4)
    return kw


if GPU:
    import h264decode_amd as H
else:
    import oracle
bad = 0
t0 = time.time()
for t in range(N if BATCH == 1 and not CONCAT else 0):
    kw = draw()
    if os.environ.get("SWEEP_VERBOSE"):  # (to find the recipe of a trial that takes the process down)
        print("trial %d recipe %s" % (t, kw), flush=True)
    try:
        s, rec, sizes = streamgen.encode(**kw)
    except RuntimeError as e:
        print("trial %d: generator refused %s (%s)" % (t, kw, e))
        continue
    W, Hc = (kw["width"] + 15) // 16 * 16, (kw["height"] + 15) // 16 * 16
    ok = True
    try:
        if GPU:
            nsl = max(1, kw.get("slices", 1)) * max(1, kw.get("slice_groups", 1))
            for x in ((str([2, 3, 4, 5, 7, 9, 13, 31, 64, 100, 512][int(rng.integers(0, 11))]),) if XR else ("256", "0")):
                os.environ["H264MI_X_WGS"] = x
                # (a field is a picture of its own for the decoder's batch limit; fed picture by picture the co-located field of a stream's first B field
                # can lie more than one call back: its motion must be kept from the start -- b_pictures)
                dec = H.Decoder(allow_unpinned_field_cabac=1, max_streams=1, max_width=W, max_height=Hc, max_frames_per_batch=kw["frames"] * (2 if kw.get("field_pics") else 1), max_slices_per_frame=nsl,
                                b_pictures=1 if (SPLIT and kw.get("field_pics") and kw.get("bframes")) else 0)
                if SPLIT and kw.get("field_pics"):  # pieces of whole pictures (the generator's sizes are per FRAME)
                    sp = H.AccessUnitSplitter(max_units_per_chunk=1)
                    aus = sp.feed(s) + sp.flush()
                    got, k = [], 0
                    while k < len(aus):
                        n = int(rng.integers(1, len(aus) - k + 1))
                        dec.decode([b"".join(aus[k:k + n])])
                        got.append(dec.read_frames(0, crop=False))
                        k += n
                    out = np.concatenate([g for g in got if g.size])
                elif SPLIT:
                    got, pos, k = [], 0, 0
                    while k < len(sizes):
                        n = int(rng.integers(1, len(sizes) - k + 1))
                        nbytes = int(sizes[k:k + n].sum())
                        dec.decode([s[pos:pos + nbytes]])
                        got.append(dec.read_frames(0, crop=False))
                        pos, k = pos + nbytes, k + n
                    out = np.concatenate([g for g in got if g.size])
                else:
                    dec.decode([s])
                    out = dec.read_frames(0, crop=False)
                dec.close()
                ok = ok and out.shape == rec.shape and np.array_equal(out, rec)
        else:
            out, _ = oracle.decode(s, crop=False)
            ok = out.shape == rec.shape and np.array_equal(out, rec)
    except Exception as e:  # noqa: BLE001
        ok = False
        print("trial %d: %s" % (t, repr(e)[:300]))
    if not ok:
        bad += 1
        print("MISMATCH trial %d: %s" % (t, kw), flush=True)
    if (t + 1) % 25 == 0:
        print("trial %d, %d mismatches, %.1fs" % (t + 1, bad, time.time() - t0), flush=True)
for t in range(N if CONCAT else 0):
    parts = []
    while len(parts) < 2:
        kw = draw()
        try:
            parts.append((kw, streamgen.encode(**kw)))
        except RuntimeError:
            pass
    stream = parts[0][1][0] + parts[1][1][0]
    dims = [((kw["width"] + 15) // 16 * 16, (kw["height"] + 15) // 16 * 16) for kw, _ in parts]
    want = [f[:w * h * 3 // 2] for (kw, g), (w, h) in zip(parts, dims) for f in g[1]]
    ok = True
    try:
        if GPU:
            for x in ("256", "0"):
                os.environ["H264MI_X_WGS"] = x
                dec = H.Decoder(allow_unpinned_field_cabac=1, max_streams=1, max_width=max(d[0] for d in dims), max_height=max(d[1] for d in dims), max_frames_per_batch=sum(kw["frames"] * (2 if kw.get("field_pics") else 1) for kw, _ in parts),
                                max_slices_per_frame=max(max(1, kw.get("slices", 1)) * max(1, kw.get("slice_groups", 1)) for kw, _ in parts))
                got = []
                if SPLIT:  # ... and in pieces of whole access units: the change may fall on a batch boundary or inside a batch
                    sizes = np.concatenate([g[2] for _, g in parts])
                    pos, k = 0, 0
                    while k < len(sizes):
                        m = int(rng.integers(1, len(sizes) - k + 1))
                        nbytes = int(sizes[k:k + m].sum())
                        dec.decode([stream[pos:pos + nbytes]])
                        got += [dec.read_frame(0, f, False) for f in range(dec.frame_count(0))]
                        pos, k = pos + nbytes, k + m
                else:
                    dec.decode([stream])
                    got = [dec.read_frame(0, f, False) for f in range(dec.frame_count(0))]
                dec.close()
                ok = ok and len(got) == len(want) and all(np.array_equal(a[:b.size], b) for a, b in zip(got, want))
        else:
            if dims[0] == dims[1]:
                out, _ = oracle.decode(stream, crop=False)
                ok = len(out) == len(want) and all(np.array_equal(a, b) for a, b in zip(out, want))
    except Exception as e:  # noqa: BLE001
        ok = False
        print("trial %d: %s" % (t, repr(e)[:300]))
    if not ok:
        bad += 1
        print("MISMATCH trial %d: %s\n  + %s" % (t, parts[0][0], parts[1][0]), flush=True)
    if (t + 1) % 25 == 0:
        print("trial %d, %d mismatches, %.1fs" % (t + 1, bad, time.time() - t0), flush=True)
for t in range(N if BATCH > 1 else 0):  # several streams per decoder
    kws, gen = [], []
    while len(gen) < BATCH:
        kw = draw()
        try:
            gen.append(streamgen.encode(**kw))
            kws.append(kw)
        except RuntimeError:
            pass
    W = max((kw["width"] + 15) // 16 * 16 for kw in kws)
    Hc = max((kw["height"] + 15) // 16 * 16 for kw in kws)
    ok = True
    try:
        for x in ("256", "0"):
            os.environ["H264MI_X_WGS"] = x
            dec = H.Decoder(allow_unpinned_field_cabac=1, max_streams=BATCH, max_width=W, max_height=Hc, max_frames_per_batch=max(kw["frames"] * (2 if kw.get("field_pics") else 1) for kw in kws),
                            max_slices_per_frame=max(max(1, kw.get("slices", 1)) * max(1, kw.get("slice_groups", 1)) for kw in kws))
            dec.decode([g[0] for g in gen])
            for i, (kw, g) in enumerate(zip(kws, gen)):
                w, h = (kw["width"] + 15) // 16 * 16, (kw["height"] + 15) // 16 * 16
                out = dec.read_frames(i, crop=False, size=w * h * 3 // 2)
                if out.shape != g[1].shape or not np.array_equal(out, g[1]):
                    ok = False
                    print("MISMATCH trial %d stream %d (x_wgs %s): %s" % (t, i, x, kw), flush=True)
            dec.close()
    except Exception as e:  # noqa: BLE001
        ok = False
        print("trial %d: %s\n  %s" % (t, repr(e)[:300], kws), flush=True)
    bad += not ok
    if (t + 1) % 25 == 0:
        print("trial %d, %d mismatches, %.1fs" % (t + 1, bad, time.time() - t0), flush=True)
print("sweep %s: %d trials, %d mismatches" % ("GPU" if GPU else "oracle", N, bad))
sys.exit(1 if bad else 0)

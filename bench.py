#!/usr/bin/env python3
"""bench.py -- headline benchmark: 1080p Main-profile CABAC decode throughput on MI355X.

Metric (BASELINE.json): "1080p Main CABAC frames/sec; aggregate Mpixels/s at 1/2/4/8 GPUs".
Workload (configs[2] x the per-GPU share of configs[4]): `--streams` independent synthetic 1080p
Main-profile CABAC streams per GPU, one IPPP GOP of `--frames` frames each (IDR + P...), QP 28,
~33 KB per P frame (streamgen recipe C3, seeds 1000 + global stream index).

A step = one pass of the GPU hot path over the whole batch (entropy decode of every slice, then
per picture inter MC + intra + deblock), with the inputs (RBSP bytes + slice/picture descriptors)
already resident in HBM: h264mi_batch_prepare() runs once, before the timed region; the timed
region calls h264mi_batch_execute() K times.  The prepare-inclusive rate (host NAL/header parse +
H2D) is reported separately as `end_to_end_fps` and is never `value`.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL); streams are independent so
there is no data-path collective -- only the closing barrier and a stats all-reduce.  Weak scaling:
per-GPU work is fixed.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


_GEN_TAG = None


def _gen_tag():
    """Identifies the generator build: cached streams are only valid for the sources that made them."""
    global _GEN_TAG
    if _GEN_TAG is None:
        import hashlib
        h = hashlib.md5()
        d = os.path.join(ROOT, "streamgen")
        for f in sorted(os.listdir(d)):
            if f.endswith((".c", ".h", ".py")):
                h.update(open(os.path.join(d, f), "rb").read())
        _GEN_TAG = h.hexdigest()[:12]
    return _GEN_TAG


def _cache_dir():
    """Scratch directory of generated streams: H264MI_BENCH_CACHE, else $XDG_CACHE_HOME/h264mi_bench, else a per-user directory under the
    temp dir.  Created with mode 0700 and used only if it belongs to this user and nobody else can write to it -- the files in it
    become the bench's inputs AND its parity reference.  Plain arrays (np.savez, loaded with allow_pickle=False), never pickles."""
    cdir = os.environ.get("H264MI_BENCH_CACHE")
    if cdir == "off":
        return None
    if not cdir:
        base = os.environ.get("XDG_CACHE_HOME")
        cdir = os.path.join(base, "h264mi_bench") if base else os.path.join(os.environ.get("TMPDIR", "/tmp"), "h264mi_bench_cache_%d" % os.getuid())
    try:
        os.makedirs(cdir, mode=0o700, exist_ok=True)
        st = os.stat(cdir)
        if st.st_uid != os.getuid() or (st.st_mode & 0o022):
            return None
    except OSError:
        return None
    return cdir


def gen_stream(args):
    """One synthetic input stream (not timed).  Generating 256 distinct 1080p GOPs costs minutes of host time, and the driver runs
    this script several times on one node (N = 1, 2, 4, 8): the streams are kept in a scratch directory (_cache_dir;
    H264MI_BENCH_CACHE=off disables) keyed by recipe, seed and the generator's source hash."""
    import hashlib
    import streamgen
    seed, frames, width, height = args[:4]
    over = args[4] if len(args) > 4 else {}
    kw = streamgen.recipe(over.get("recipe", "C3"), frames=frames, idr_period=frames, seed=seed, width=width, height=height)
    kw.update({k: v for k, v in over.items() if k != "recipe"})
    if os.environ.get("H264MI_BENCH_DBF"):  # experiments only: disable_deblocking_filter_idc of the synthetic streams
        kw["deblock_idc"] = int(os.environ["H264MI_BENCH_DBF"])
    if os.environ.get("H264MI_BENCH_INTRAP"):  # experiments only: share of intra macroblocks in P pictures (per mille)
        kw["intra_in_p_permille"] = int(os.environ["H264MI_BENCH_INTRAP"])
    cdir = _cache_dir()
    path = None
    if cdir:
        key = hashlib.md5(repr(sorted(kw.items())).encode()).hexdigest()[:16]
        path = os.path.join(cdir, "%s_%s.npz" % (_gen_tag(), key))
        try:
            with np.load(path, allow_pickle=False) as z:
                rec = z["rec"]
                return (z["stream"].tobytes(), [bytes(r) for r in rec] if rec.shape[1] == 16 else rec, z["sizes"])
        except Exception:
            pass
    s, rec, sizes = streamgen.encode(want_recon=True, **kw)
    # keep only what the parity gate needs (host memory: 256 x 30 x 3.1 MB otherwise): all frames of the first two
    # streams, the MD5 of every frame of the others
    if seed % 1000 >= 2:
        rec = [hashlib.md5(f.tobytes()).digest() for f in rec]
    out = (s, rec, sizes)
    if path:
        try:
            tmp = "%s.%d.tmp.npz" % (path, os.getpid())
            np.savez(tmp, stream=np.frombuffer(s, dtype=np.uint8), sizes=np.asarray(sizes),
                     rec=np.frombuffer(b"".join(rec), dtype=np.uint8).reshape(len(rec), 16) if isinstance(rec, list) else rec)
            os.replace(tmp, path)
        except Exception:
            pass  # a scratch directory that cannot be written is not an error
    return out


GEN_CORE_SECONDS_1080P_GOP = 29.0  # measured: one 1080p GOP-30 C3 stream on one host core of the GPU node (round 3: 256 streams, 62 threads, 121 s)


def generation_threads(n_distinct, world):
    """Generator threads of one rank: its share of the host's cores (every rank of an N-GPU run generates on the same host)."""
    return max(1, min(n_distinct, (os.cpu_count() or 8) // max(1, world) - (2 if world == 1 else 0)))


def planned_generation_seconds(n_distinct, world, cores=None, frames=30):
    """Wall-clock estimate of a rank's input generation (tests/test_bench_contract.py holds the N = 8 plan against the time limit)."""
    cores = cores or (os.cpu_count() or 8)
    threads = max(1, min(n_distinct, cores // max(1, world) - (2 if world == 1 else 0)))
    return -(-n_distinct // threads) * GEN_CORE_SECONDS_1080P_GOP * frames / 30.0


def timed_fps(dec, streams, n_frames, steps, warmup=1):
    """frames/s of `steps` passes over an already prepared batch (inputs resident in HBM), and the prepare-inclusive rate."""
    import torch
    dec.prepare(streams)
    for _ in range(warmup):
        dec.execute()
    dec.sync()
    torch.cuda.synchronize()
    def run(k):
        t0 = time.perf_counter()
        for _ in range(k):
            dec.execute()
        dec.sync()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    total = run(steps)
    dt = total / steps
    # `fps` starts from an empty pipeline: with few streams the entropy stage of the first pass (one I slice: 205 ms) is a
    # visible share of a short run.  `steady_fps` is the marginal rate: the time 2k passes take minus the time k take.
    marginal = max(run(2 * steps) - total, 1e-9) / steps
    t0 = time.perf_counter()
    dec.decode(streams)
    torch.cuda.synchronize()
    e2e = time.perf_counter() - t0
    used, cap = dec.coef_pool()
    return {"fps": round(n_frames / dt, 2), "ms_per_step": round(dt * 1e3, 3), "steady_fps": round(n_frames / marginal, 2), "end_to_end_fps": round(n_frames / e2e, 2),
            "coef_pool_used": round(used / cap, 3)}


def extra_configs(H, streams, F, W, Hc, device, args):
    """The contract's own configurations next to `value`: C5's per-GPU share -- 32 distinct streams, one GOP
    each and 8 GOPs each -- and C3 proper: ONE stream of 300 frames.  Longer streams are concatenations of the 30-frame
    GOP streams already generated (each starts with SPS + PPS + IDR, so the result is a valid multi-GOP stream).
    A configuration that fails reports {"error": ...} under its key: the headline measurement has been taken by then and its
    line must still be printed."""
    import streamgen
    out = {}
    S = len(streams)

    def guarded(key, fn):
        try:
            out[key] = fn()
        except Exception as e:  # noqa: BLE001 -- reported in the line, not swallowed
            out[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    def c5_share():
        n32 = min(32, S)
        gops = max(1, min(8, S // n32))
        dec = H.Decoder(max_streams=n32, max_width=W, max_height=Hc, max_frames_per_batch=F * gops, max_slices_per_frame=1, device=device,
                        max_bitstream_bytes=int(sum(len(s) for s in streams[:n32 * gops]) * 1.1) + (1 << 20))
        try:
            r = timed_fps(dec, streams[:n32], n32 * F, steps=max(2, args.steps))
            r["workload"] = "%d distinct streams x %d frames" % (n32, F)
            res = {"one_gop": r}
            if gops > 1:
                deep = [b"".join(streams[i * gops:(i + 1) * gops]) for i in range(n32)]
                r = timed_fps(dec, deep, n32 * F * gops, steps=4)
                r["workload"] = "%d distinct streams x %d frames (%d GOPs each)" % (n32, F * gops, gops)
                res["deep"] = r
            return res
        finally:
            dec.close()

    def single_stream():
        g1 = max(1, min(10, S))
        one = b"".join(streams[:g1])
        dec = H.Decoder(max_streams=1, max_width=W, max_height=Hc, max_frames_per_batch=F * g1, max_slices_per_frame=1, device=device,
                        max_bitstream_bytes=int(len(one) * 1.1) + (1 << 20))
        try:
            r = timed_fps(dec, [one], F * g1, steps=6)  # (the first pass's entropy stage is not hidden behind a previous pass: amortise it)
            r["workload"] = "C3: 1 stream x %d frames (%d GOPs)" % (F * g1, g1)
            return r
        finally:
            dec.close()

    def fractional_motion():
        # K4 on FRACTIONAL motion (SURVEY 8d C3 asks for quarter-sample vectors; the default scene moves whole samples): the same
        # recipe with the scene moving (2.75, -1.5) samples per frame, 8 distinct streams replicated to the batch size of `value`
        nf = 8
        with ThreadPoolExecutor(max_workers=max(1, min(nf, (os.cpu_count() or 8) - 1))) as ex:
            fgen = list(ex.map(lambda sd: streamgen.encode(want_recon=True, **streamgen.recipe("C3", frames=F, idr_period=F, seed=sd, width=W, height=args.height,
                                                                                             motion_x4=11, motion_y4=-6)), range(2000, 2000 + nf)))
        fstreams = [fgen[i % nf][0] for i in range(S)]
        dec = H.Decoder(max_streams=S, max_width=W, max_height=Hc, max_frames_per_batch=F, max_slices_per_frame=1, device=device,
                        max_bitstream_bytes=int(sum(len(s) for s in fstreams) * 1.1) + (1 << 20))
        try:
            r = timed_fps(dec, fstreams, S * F, steps=max(2, min(args.steps, 5)))
            got = dec.read_frames(S - 1, crop=False)
            r["parity"] = "bit-exact vs streamgen recon (stream %d, all frames)" % (S - 1) if np.array_equal(got, fgen[(S - 1) % nf][1]) else "MISMATCH"
            dec.set_profiling(True)
            dec.execute()
            dec.sync()
            lt = np.array(dec.launch_times_ms("inter"))
            dec.set_profiling(False)
            fsz = W * Hc * 3 // 2
            r["k_inter"] = {"ms": round(float(lt.mean()), 4), "GB/s": round(2.0 * fsz * S / (float(lt.mean()) * 1e-3) / 1e9, 2),
                            "frac": round(2.0 * fsz * S / (float(lt.mean()) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
            r["workload"] = "%d streams (%d distinct) x %d frames, scene motion (2.75, -1.5) samples per frame: fractional vectors are the rule" % (S, nf, F)
            return r
        finally:
            dec.close()

    def b_pictures():
        # B pictures: the same recipe coded I B B P ... (two B pictures between the anchors, three reference
        # frames, spatial direct), through k_entropy_b / k_inter_b.  16 distinct streams, each used twice.
        nb = 16
        with ThreadPoolExecutor(max_workers=max(1, min(nb, (os.cpu_count() or 8) - 1))) as ex:
            gen = list(ex.map(lambda sd: streamgen.encode(want_recon=True, **dict(streamgen.recipe("C3", frames=F, idr_period=F, seed=sd, width=W, height=args.height),
                                                                                      bframes=2, num_ref_frames=3, bskip_permille=300)), range(3000, 3000 + nb)))
        bstreams = [g[0] for g in gen] * 2
        dec = H.Decoder(max_streams=len(bstreams), max_width=W, max_height=Hc, max_frames_per_batch=F, max_slices_per_frame=1, device=device,
                        max_bitstream_bytes=int(sum(len(s) for s in bstreams) * 1.1) + (1 << 20))
        try:
            r = timed_fps(dec, bstreams, len(bstreams) * F, steps=max(2, args.steps))
            got = dec.read_frames(nb + 1, crop=False)
            r["parity"] = "bit-exact vs streamgen recon (stream %d, all frames)" % (nb + 1) if np.array_equal(got, gen[1][1]) else "MISMATCH"
            r["workload"] = "%d streams (%d distinct) x %d frames, I B B P coding order, 3 reference frames" % (len(bstreams), nb, F)
            return r
        finally:
            dec.close()

    def kernel_ms(dec):
        dec.set_profiling(True)
        dec.execute()
        dec.sync()
        kt = dec.kernel_times_ms()
        lt = {k: dec.launch_times_ms(k) for k in ("inter", "intra", "deblock")}
        dec.set_profiling(False)
        return kt, lt

    def c2_720p_cavlc_intra():
        # BASELINE configs[1]: 720p Baseline CAVLC, every frame an IDR picture: per-MB dequant + 4x4 IDCT + intra prediction is the whole reconstruction
        n, fr = 32, 60
        gen = args.gen_c2
        cs = [gen[i % 8][0] for i in range(n)]
        # (all-intra content at QP 28 fills ~9 residual blocks per macroblock: above the default pool of 8 -- every pass would be repeated with the whole
        # allocation (h264mi_batch_sync) --, so this decoder is created with room for 16)
        dec = H.Decoder(max_streams=n, max_width=1280, max_height=720, max_frames_per_batch=fr, max_slices_per_frame=1, device=device,
                        max_bitstream_bytes=int(sum(len(x) for x in cs) * 1.1) + (1 << 20), coef_blocks_per_mb=16)
        try:
            r = timed_fps(dec, cs, n * fr, steps=max(2, min(args.steps, 5)))
            rec = gen[(n - 1) % 8][1]
            got = dec.read_frame(n - 1, fr - 1, crop=False)[:1280 * 720 * 3 // 2]
            import hashlib
            same = np.array_equal(got, rec[fr - 1]) if isinstance(rec[fr - 1], np.ndarray) else hashlib.md5(got.tobytes()).digest() == rec[fr - 1]
            r["parity"] = "bit-exact vs streamgen recon (stream %d, last frame)" % (n - 1) if same else "MISMATCH"
            kt, lt = kernel_ms(dec)
            ims = float(np.mean(lt["intra"]))
            fb = 1280 * 720 * 3 // 2
            ppl = n * fr // max(1, len(lt["intra"]))  # (all-intra streams: no picture waits for another one -- the whole batch is one launch)
            r["k_intra"] = {"ms_per_launch": round(ims, 4), "pictures_per_launch": ppl, "GB/s": round(fb * ppl / (ims * 1e-3) / 1e9, 2),
                            "frac": round(fb * ppl / (ims * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": fb * ppl}
            r["kernel_ms_per_step"] = {k: round(v, 3) for k, v in kt.items()}
            r["workload"] = "C2: %d streams (8 distinct) x %d frames, 1280x720 Baseline CAVLC, all IDR" % (n, fr)
            return r
        finally:
            dec.close()

    def c4_4k_high_8slices():
        # BASELINE configs[3]: 4K High profile, 8x8 transform, 8 slices per picture: slice-parallel CABAC (one slice per wavefront)
        n, fr = 8, 30
        gen = args.gen_c4
        cs = [gen[i % 2][0] for i in range(n)]
        dec = H.Decoder(max_streams=n, max_width=3840, max_height=2160, max_frames_per_batch=fr, max_slices_per_frame=8, device=device,
                        max_bitstream_bytes=int(sum(len(x) for x in cs) * 1.1) + (1 << 20))
        try:
            r = timed_fps(dec, cs, n * fr, steps=max(2, min(args.steps, 5)))
            rec = gen[(n - 1) % 2][1]
            got = dec.read_frame(n - 1, fr - 1, crop=False)[:3840 * 2160 * 3 // 2]
            import hashlib
            same = np.array_equal(got, rec[fr - 1]) if isinstance(rec[fr - 1], np.ndarray) else hashlib.md5(got.tobytes()).digest() == rec[fr - 1]
            r["parity"] = "bit-exact vs streamgen recon (stream %d, last frame)" % (n - 1) if same else "MISMATCH"
            kt, lt = kernel_ms(dec)
            r["kernel_ms_per_step"] = {k: round(v, 3) for k, v in kt.items()}
            r["entropy"] = {"kernel_ms": round(kt["entropy"], 3), "slices_in_flight": n * fr * 8,
                            "bits_per_s": round(sum(len(x) for x in cs) * 8 / (kt["entropy"] * 1e-3), 0)}
            r["workload"] = "C4: %d streams (2 distinct) x %d frames, 3840x2160 High CABAC, 8x8 transform, 8 slices per picture" % (n, fr)
            return r
        finally:
            dec.close()

    guarded("c5_share", c5_share)
    guarded("c2_720p_cavlc_intra", c2_720p_cavlc_intra)
    guarded("c4_4k_high_8slices", c4_4k_high_8slices)
    guarded("single_stream", single_stream)
    guarded("fractional_motion", fractional_motion)
    guarded("b_pictures", b_pictures)
    return out


def strong_share(H, torch, dist, streams, F, W, Hc, device, args, rank, world, total=256, gops=8, steps=2):
    """C5 as ONE fixed job on the N > 1 line (SURVEY 8e "256 streams total"): `total` streams of `gops` GOPs each, stream s on
    rank s mod N, built from the GOP streams the rank has already generated; deep batches (several GOPs of every stream per
    batch) and pipelined ingest as in run_strong.  Returns rank 0's dict (None elsewhere)."""
    from h264decode_amd.dist import allreduce_stats
    S = len([s for s in range(total) if s % world == rank])
    nd = len(streams)
    D = max(1, min(gops, 256 // max(S, 1)))
    while gops % D:
        D -= 1
    nbatch = gops // D
    batches = [[b"".join(streams[(k + S * (b * D + j)) % nd] for j in range(D)) for k in range(S)] for b in range(nbatch)]
    dec = H.Decoder(max_streams=max(S, 1), max_width=W, max_height=Hc, max_frames_per_batch=F * D, max_slices_per_frame=1, device=device,
                    max_bitstream_bytes=int(max(sum(len(x) for x in bt) for bt in batches) * 1.1) + (1 << 20))
    seq = [b for _ in range(1 + steps) for b in batches]
    dec.prepare(seq[0])
    for k in range(nbatch):  # one warm-up step through the same pipeline
        dec.execute()
        dec.prepare(seq[k + 1])
    dec.sync()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(nbatch, len(seq)):
        dec.execute()
        if k + 1 < len(seq):
            dec.prepare(seq[k + 1])
    dec.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    red = allreduce_stats({"ranks": 1, "frames": S * gops * F * steps, "seconds": time.perf_counter() - t0}, device="cuda" if (dist and dist.get_backend() == "nccl") else None)
    dec.close()
    if rank != 0:
        return None
    return {"fps": round(red["frames"] / red["seconds"], 2), "ms_per_step": round(red["seconds"] / steps * 1e3, 3), "scaling": "strong", "ranks_seen": int(red.get("ranks", 1)),
            "workload": "%d streams x %d GOPs in total, stream s on rank s mod %d; rank 0: %d streams, %d batch(es) of %d GOP(s); ingest pipelined inside the timed region"
                        % (total, gops, world, S, nbatch, D)}


def run_strong(args, H, torch, dist, rank, world, local_rank):
    """STRONG scaling: a fixed job -- `--total-streams` streams of `--gops` GOPs each -- sharded over the GPUs
    by longest-processing-time-first on the streams' byte counts (dist.shard_streams_lpt).  A GPU with few streams gets
    deep batches (more GOPs of each stream per batch) so that its batch still holds ~256 stream-GOPs and fills the chip;
    with several batches per step ingest is pipelined (prepare(k + 1) overlaps execute(k)), so here the host parse and the
    H2D copies are inside the timed region."""
    from h264decode_amd.dist import shard_streams_lpt, allreduce_stats
    T, G, F = args.total_streams, args.gops, args.frames
    W, Hc = (args.width + 15) // 16 * 16, (args.height + 15) // 16 * 16
    nd = max(1, min(args.distinct or T, T))
    # every rank generates a round-robin share of the distinct GOP streams, sizes are exchanged, LPT decides the owners
    mine = [i for i in range(nd) if i % world == rank]
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=max(1, min(len(mine), (os.cpu_count() or 8) // max(1, min(world, 8)) - 1))) as ex:
        gen = dict(zip(mine, ex.map(gen_stream, [(1000 + i, F, args.width, args.height) for i in mine])))
    sizes = torch.zeros(nd, dtype=torch.int64, device="cuda" if (not dist or dist.get_backend() == "nccl") else "cpu")
    for i, g in gen.items():
        sizes[i] = len(g[0])
    if dist:
        dist.all_reduce(sizes)
    sizes = [int(x) for x in sizes.tolist()]
    costs = [sum(sizes[(s + 7 * j) % nd] for j in range(G)) for s in range(T)]  # stream s = GOP streams s, s+7, s+14, ... (mod nd)
    shard = shard_streams_lpt(costs, world, rank)
    need = sorted({(s + 7 * j) % nd for s in shard for j in range(G)} - set(gen))
    with ThreadPoolExecutor(max_workers=max(1, min(len(need) or 1, (os.cpu_count() or 8) // max(1, min(world, 8)) - 1))) as ex:
        gen.update(zip(need, ex.map(gen_stream, [(1000 + i, F, args.width, args.height) for i in need])))
    gen_s = time.time() - t0
    S = len(shard)
    D = max(1, min(G, 256 // max(S, 1)))  # GOPs per batch
    while G % D:
        D -= 1
    nbatch = G // D
    batches = [[b"".join(gen[(s + 7 * (b * D + j)) % nd][0] for j in range(D)) for s in shard] for b in range(nbatch)]
    dec = H.Decoder(max_streams=max(S, 1), max_width=W, max_height=Hc, max_frames_per_batch=F * D, max_slices_per_frame=1, device=local_rank,
                    max_bitstream_bytes=int(max(sum(len(x) for x in bt) for bt in batches) * 1.1) + (1 << 20))
    # parity gate: the first GOP of every stream of the first batch against the generator's reconstruction
    import hashlib
    fsz = W * Hc * 3 // 2
    dec.decode(batches[0])
    for k, s in enumerate(shard):
        rec = gen[s % nd][1]
        out = dec.read_frame(k, F - 1, crop=False)[:fsz]
        same = np.array_equal(out, rec[F - 1]) if isinstance(rec[F - 1], np.ndarray) else hashlib.md5(out.tobytes()).digest() == rec[F - 1]
        if not same:
            raise SystemExit("PARITY FAILURE: stream %d differs from the reference reconstruction" % s)
    seq = [b for _ in range(args.warmup + args.steps) for b in batches]
    dec.prepare(seq[0])
    k0 = args.warmup * nbatch
    for k in range(k0):  # warm-up steps, same pipeline
        dec.execute()
        dec.prepare(seq[k + 1])
    dec.sync()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for k in range(k0, len(seq)):
        dec.execute()
        if k + 1 < len(seq):
            dec.prepare(seq[k + 1])
    dec.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    red = allreduce_stats({"ranks": 1, "frames": S * G * F * args.steps, "pixels": S * G * F * args.steps * args.width * args.height,
                           "bytes_in": sum(len(x) for bt in batches for x in bt) * args.steps, "seconds": elapsed}, device="cuda" if (dist and dist.get_backend() == "nccl") else None)
    dec.close()
    if rank == 0:
        fps = red["frames"] / red["seconds"]
        print(json.dumps({
            "metric": "1080p Main CABAC frames/sec", "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "ranks_seen": int(red.get("ranks", 1)), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(red["seconds"] / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%dx%d Main CABAC IPPP GOP-%d: %d streams x %d GOPs in total, sharded over %d GPU(s) by LPT on stream bytes; rank 0: %d streams, "
                                   "%d batch(es) of %d GOP(s) per step, ingest pipelined inside the timed region" % (args.width, args.height, F, T, G, world, S, nbatch, D),
                       "frames_per_step": T * G * F, "parallelism": "streams sharded, no collective"},
            "mpixels_per_s": round(fps * args.width * args.height / 1e6, 1), "roofline": None, "cpu_baseline": None,
            "bytes_in_per_step": red["bytes_in"] / args.steps, "stream_gen_s": round(gen_s, 1),
            "parity": "bit-exact vs streamgen recon (last frame of the first GOP of every stream of rank 0)"}))
    if dist:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=256, help="independent streams per GPU")
    ap.add_argument("--frames", type=int, default=30, help="frames per stream per step (one GOP)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic streams generated per GPU (0 = all of them, SURVEY 8d C5: seeds 1000 + global stream index); fewer are replicated")
    ap.add_argument("--total-streams", type=int, default=0, help="STRONG scaling: this many streams in total, sharded over the GPUs (LPT on stream bytes); "
                                                                  "each stream is --gops GOPs long and a step decodes all of it, in batches deep enough to fill a GPU")
    ap.add_argument("--gops", type=int, default=8, help="GOPs per stream and step in the strong-scaling mode")
    ap.add_argument("--no-extra", action="store_true", help="skip the c5_share (32 streams per GPU) and single_stream (C3: 1 stream x 300 frames) measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--serialized", action="store_true", help="profiling aid: the timed passes run in the library's profiling mode (stages back to back on one stream), so that "
                                                              "a kernel trace of this command shows uncontended kernel durations -- the quantity `roofline.launch_ms` is; `value` is then not the contract's")
    ap.add_argument("--max-ref-frames", type=int, default=0, help="h264mi_config.max_ref_frames of the bench decoder (0 = the default of 16 reference slots per stream; "
                    "the synthetic streams use 1: 4 would save 9.6 GB of the 151 GB)")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous check only: every rank joins the process group, the closing all-reduce runs and rank 0 "
                                                           "prints n_gpus / ranks_seen; nothing is decoded (works without a GPU over gloo)")
    args = ap.parse_args()

    # `python bench.py --gpus N` starts its own N ranks (one process per GPU).  This process has not imported torch or touched
    # HIP yet and never will: the ranks run as a CHILD process (torch.distributed.run), whose single JSON line and exit code
    # are relayed.  Under a launcher (WORLD_SIZE set) this branch is skipped and --gpus is checked against the world size.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        return subprocess.call(cmd)

    import torch
    import h264decode_amd as H

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if args.dry_run:
        import torch.distributed as dist
        from h264decode_amd.dist import allreduce_stats
        if world > 1:
            dist.init_process_group(os.environ.get("H264MI_BENCH_BACKEND", "nccl" if torch.cuda.is_available() else "gloo"))
        red = allreduce_stats({"frames": 1.0, "ranks": 1.0, "seconds": float(rank)})
        if rank == 0:
            nd_plan = max(1, min(args.distinct or (args.streams if world == 1 else 32), args.streams))
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": int(red["ranks"]), "slowest_rank_seconds": red["seconds"],
                              "distinct_streams_per_rank": nd_plan, "generator_threads_per_rank": generation_threads(nd_plan, world),
                              "planned_generation_s": round(planned_generation_seconds(nd_plan, world, frames=args.frames), 1)}))
        if world > 1:
            dist.destroy_process_group()
        return 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decode path has no CPU fallback")
    # rehearsal hooks (one-GPU box): H264MI_BENCH_DEVICE pins every rank to one device, H264MI_BENCH_BACKEND=gloo replaces RCCL
    if os.environ.get("H264MI_BENCH_DEVICE"):
        local_rank = int(os.environ["H264MI_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("H264MI_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if args.total_streams > 0:
        return run_strong(args, H, torch, dist, rank, world, local_rank)
    S, F = args.streams, args.frames
    # Distinct streams per rank: all of them on one GPU.  With N > 1 every rank generates its own inputs on the SAME host: 8 x 256
    # distinct 1080p GOPs are ~16 core-hours of generator time, more than the bench's time limit on any host -- so a rank generates 32 distinct
    # streams by default (each used 8 times; --distinct overrides) with its share of the host's cores, and the line says so (config.workload).
    nd = max(1, min(args.distinct or (S if world == 1 else 32), S))
    gen_threads = generation_threads(nd, world)
    # ---- synthetic inputs (not timed) ----
    t0 = time.time()
    seeds = [1000 + rank * S + i for i in range(nd)]
    # (the inputs of the contract's other configurations -- extra_configs: C2, C4 -- are generated by the same pool, longest jobs first)
    jobs_x = [] if (args.no_extra or world > 1) else [(4000 + i, 30, 3840, 2160, {"recipe": "C4", "idr_period": 30}) for i in range(2)] + \
        [(2000 + i, 60, 1280, 720, {"recipe": "C2", "idr_period": 1}) for i in range(8)]
    with ThreadPoolExecutor(max_workers=gen_threads) as ex:
        fx = [ex.submit(gen_stream, j) for j in jobs_x]
        gen = list(ex.map(gen_stream, [(sd, F, args.width, args.height) for sd in seeds]))
        args.gen_c4, args.gen_c2 = [f.result() for f in fx[:2]], [f.result() for f in fx[2:]]
    streams = [gen[i % nd][0] for i in range(S)]
    gen_s = time.time() - t0
    W, Hc = (args.width + 15) // 16 * 16, (args.height + 15) // 16 * 16
    fsz = W * Hc * 3 // 2
    bytes_per_frame = float(np.mean([len(g[0]) for g in gen])) / F

    hip_stream = torch.cuda.current_stream().cuda_stream
    dec = H.Decoder(max_streams=S, max_width=W, max_height=Hc, max_frames_per_batch=F, max_slices_per_frame=1, device=local_rank,
                    max_bitstream_bytes=int(sum(len(s) for s in streams) * 1.1) + (1 << 20), hip_stream=hip_stream, max_ref_frames=args.max_ref_frames)
    hbm_bytes = dec.device_bytes()
    # ---- stage 1: inputs -> HBM (not timed) ----
    tp = time.time()
    info = dec.prepare(streams)
    torch.cuda.synchronize()
    prepare_s = time.time() - tp
    assert info.n_frames == S * F, (info.n_frames, S * F)

    # ---- parity gate before any timing: GPU output == generator reconstruction, bit for bit ----
    parity = "skipped"
    if not args.no_parity:
        dec.execute()
        dec.sync()
        import hashlib
        for si in range(S):
            rec = gen[si % nd][1]
            frames = range(F) if (si < 2 or si % 16 == 0) else [F - 1]
            for f in frames:
                out = dec.read_frame(si, f, crop=False)[:fsz]
                same = np.array_equal(out, rec[f]) if isinstance(rec[f], np.ndarray) else hashlib.md5(out.tobytes()).digest() == rec[f]
                if not same:
                    raise SystemExit("PARITY FAILURE: stream %d frame %d differs from the reference reconstruction" % (si, f))
        parity = "bit-exact vs streamgen recon (every 16th stream all frames, others last frame)"

    # ---- timed region: K passes enqueued back to back; inside the library the entropy kernels of
    # pass n+1 (own HIP stream, second MbRec/coefficient buffer set) overlap reconstruction of pass n ----
    dec.set_profiling(bool(args.serialized))
    for _ in range(args.warmup):
        dec.execute()
    dec.sync()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        dec.execute()
    dec.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    # closing reduction over the ranks (no collective anywhere in the decode path): frames summed, time = the slowest rank
    from h264decode_amd.dist import allreduce_stats
    red = allreduce_stats({"ranks": 1, "frames": S * F * args.steps, "pixels": S * F * args.steps * args.width * args.height,
                           "bytes_in": sum(len(x) for x in streams) * args.steps, "seconds": elapsed}, device="cuda" if (dist and dist.get_backend() == "nccl") else None)
    gen_s = allreduce_stats({"seconds": gen_s}, device="cuda" if (dist and dist.get_backend() == "nccl") else None)["seconds"]  # the slowest rank's
    elapsed, total_frames = red["seconds"], red["frames"]
    ranks_seen = int(red.get("ranks", 1))
    fps = total_frames / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # ---- per-kernel durations: HIP events on the launch stream around every launch of 2 more passes
    # (profiling mode runs the stages back to back on one stream so that intervals are per kernel) ----
    dec.set_profiling(True)
    ktimes, ltimes = [], []
    for _ in range(2):
        dec.execute()
        dec.sync()
        ktimes.append(dec.kernel_times_ms())
        ltimes.append({k: dec.launch_times_ms(k) for k in ("inter", "intra", "deblock")})
    dec.set_profiling(False)

    # ---- K6: crop + tight I420 pack of the whole batch in one launch (2 x display bytes per frame) ----
    k_pack = None
    try:
        disp = args.width * args.height * 3 // 2
        pbuf = torch.empty(S * F * disp, dtype=torch.uint8, device="cuda")
        dec.pack_batch(pbuf.data_ptr(), pbuf.numel())
        dec.sync()
        torch.cuda.synchronize()
        # wall clock around launch + sync of the decoder's own stream (a torch.cuda.Event would sit on torch's current
        # stream and not see the kernel); the launch overhead is microseconds against a 9 ms kernel
        t0 = time.perf_counter()
        for _ in range(3):
            dec.pack_batch(pbuf.data_ptr(), pbuf.numel())
        dec.sync()
        pms = (time.perf_counter() - t0) / 3 * 1e3
        k_pack = {"ms": round(pms, 3), "GB/s": round(2.0 * S * F * disp / (pms * 1e-3) / 1e9, 1), "frac": round(2.0 * S * F * disp / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                  "note": "%d frames in one launch; includes the descriptor-table upload" % (S * F)}
        # `value` with K6 in the loop: every step ends with the crop + pack launch of the whole batch into a resident buffer (the output side of the
        # path: what a consumer on the device reads).  Same barrier / synchronize bracket as `value`.
        for _ in range(args.warmup):
            dec.execute()
            dec.pack_batch(pbuf.data_ptr(), pbuf.numel())
        dec.sync()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dec.execute()
            dec.pack_batch(pbuf.data_ptr(), pbuf.numel())
        dec.sync()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        wp = allreduce_stats({"frames": S * F * args.steps, "seconds": time.perf_counter() - t0}, device="cuda" if (dist and dist.get_backend() == "nccl") else None)
        k_pack["value_with_pack"] = round(wp["frames"] / wp["seconds"], 2)
        del pbuf
    except RuntimeError as e:  # not enough free HBM for the packed copy of the whole batch
        if dist:  # (the ranks must still meet at the collectives of the branch they did not all take)
            raise
        k_pack = {"skipped": str(e)[:80]}

    # ---- end-to-end rate including host parse + H2D (reported, never `value`) ----
    te = time.perf_counter()
    dec.decode(streams)
    torch.cuda.synchronize()
    e2e_s = time.perf_counter() - te

    # ---- the same with ingest in the loop, pipelined: h264mi_batch_prepare(k + 1) -- host NAL / header parsing, DPB
    # bookkeeping, H2D into the second staging set -- runs while batch k executes (h264/server.go:144-145's endless loop).
    # Reported next to `value`; `value` itself keeps the contract's "inputs already resident in HBM".
    dec.prepare(streams)
    torch.cuda.synchronize()
    tpipe = time.perf_counter()
    for k in range(args.steps):
        dec.execute()
        if k + 1 < args.steps:
            dec.prepare(streams)
    dec.sync()
    torch.cuda.synchronize()
    pipelined_fps = S * F * args.steps / (time.perf_counter() - tpipe)
    pool_used, pool_cap = dec.coef_pool()

    dec.close()
    del dec
    extra = {}
    if not args.no_extra and world == 1:
        extra = extra_configs(H, streams, F, W, Hc, local_rank, args)
    elif not args.no_extra and S >= 256 // world:  # N > 1: C5 as a fixed 256-stream job next to the weak-scaling `value`
        c5 = strong_share(H, torch, dist, streams, F, W, Hc, local_rank, args, rank, world)
        if c5:
            extra = {"c5_strong": c5}

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    kt = {k: float(np.mean([x[k] for x in ktimes])) for k in ktimes[0]}
    F_bytes = fsz  # bytes of one coded 4:2:0 frame
    # Per-launch durations (HIP events around every launch).  Launch k of a pixel kernel handles picture k of every stream:
    # launch 0 is the IDR picture of the GOP.  k_intra is priced on that launch ONLY -- in the P-picture launches it
    # touches a few per cent of the macroblocks, so averaging over all launches would inflate its rate.
    lt = {k: np.mean([x[k] for x in ltimes], axis=0) for k in ltimes[0]}
    intra_I_ms = float(lt["intra"][0])
    # algorithmic bytes per launch: inter 2F, intra F (all-intra launch), deblock 2F -- x S frames per launch
    per_launch = {
        "k_inter": (float(np.mean(lt["inter"])), 2.0 * F_bytes * S),
        "k_deblock": (float(np.mean(lt["deblock"])), 2.0 * F_bytes * S),
        "k_intra": (intra_I_ms, 1.0 * F_bytes * S),
    }
    dom = max(("k_inter", "k_deblock"), key=lambda k: kt["inter" if k == "k_inter" else "deblock"])
    dur_ms, alg_bytes = per_launch[dom]
    achieved = alg_bytes / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
    # HBM traffic of the dominant kernel comes from a separate rocprofv3 --pmc pass (counters cannot be
    # collected from inside this process): profiles/pmc_traffic.json, written by tools/pmc_traffic.sh
    # for the same workload, holds corrected bytes per launch; null when it does not match this run.
    traffic = None
    try:
        pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if pt.get("streams") == S and pt.get("frames") == F and pt.get("width") == args.width and dom in pt.get("bytes_per_launch", {}):
            traffic = pt["bytes_per_launch"][dom]
    except (OSError, ValueError):
        pass
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "launch_ms": round(dur_ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
                "all_kernels_ms_per_step": {k: round(v, 3) for k, v in kt.items()},
                "per_launch": {k: {"ms": round(v[0], 4), "GB/s": round(v[1] / (v[0] * 1e-3) / 1e9, 2) if v[0] > 0 else 0.0,
                                   "frac": round(v[1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if v[0] > 0 else 0.0} for k, v in per_launch.items()},
                "k_intra_note": "I-picture launch only (launch 0 of the GOP); its P-picture launches take %.3f ms each" % float(np.mean(lt["intra"][1:])) if len(lt["intra"]) > 1 else ""}

    cpu_baseline, bins_per_frame = None, None
    if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
        import oracle
        sample = streams[0]
        sinfo = oracle.probe(sample)  # sizing pass outside the timed region: each timed call is exactly one decode
        reps, tcpu = 0, 0.0
        while tcpu < 10.0 and reps < 8:
            t1 = time.perf_counter()
            oracle.decode(sample, crop=False, info=sinfo)
            tcpu += time.perf_counter() - t1
            reps += 1
        cpu_baseline = {"value": round(reps * F / tcpu, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                        "sample": "%d x stream 0 (%d frames 1080p Main CABAC, %.0f KB) decoded by oracle/ (scalar C restatement; the Go reference "
                                  "cannot be built and produces no pixels)" % (reps, F, len(sample) / 1e3)}
        # one independent stream per host core (ctypes releases the GIL inside the C decoder)
        ncore = max(1, min(len(os.sched_getaffinity(0)), 64))
        t1 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=ncore) as ex:
            list(ex.map(lambda i: oracle.decode(streams[i % S], crop=False, info=sinfo)[0].shape, range(ncore)))
        tmt = time.perf_counter() - t1
        cpu_baseline["all_cores"] = {"value": round(ncore * F / tmt, 2), "unit": "frames/s", "cores": ncore,
                                     "sample": "%d streams (one per core) x %d frames" % (ncore, F)}
        bins_per_frame = float(sinfo.n_bins) / max(1, int(sinfo.n_frames))  # the checker's count of the CABAC bins of stream 0 (bins/s beside bits/s)

    out = {
        "metric": "1080p Main CABAC frames/sec",
        "value": round(fps, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "ranks_seen": ranks_seen,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "value_with_pack": (k_pack or {}).get("value_with_pack"),
        "value_note": "`value`: K passes of h264mi_batch_execute over a batch resident in HBM (entropy decode -> reconstruction -> deblocked frames in the frame pool); "
                      "`value_with_pack` adds the K6 crop + pack launch of the whole batch to every step; host parse + H2D are in `end_to_end_fps` / `pipelined_ingest_fps`",
        "config": {"workload": "%dx%d Main CABAC IPPP GOP-%d, %d independent streams per GPU (%d distinct%s), QP 28, 1 slice/frame, 1 ref"
                               % (args.width, args.height, F, S, nd, "" if nd == S else ", each used %d times: %d ranks generate their inputs on one host" % (-(-S // nd), world)),
                   "frames_per_step_per_gpu": S * F, "bytes_per_frame": round(bytes_per_frame, 1), "parallelism": "streams sharded, no collective"},
        "mpixels_per_s": round(fps * args.width * args.height / 1e6, 1),
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
        "entropy_stage": {"bits_per_s": round(bytes_per_frame * 8 * S * F / (kt["entropy"] * 1e-3), 0), "slices_in_flight": S * F,
                          "bins_per_s": round(bins_per_frame * S * F / (kt["entropy"] * 1e-3), 0) if bins_per_frame else None,
                          "bins_per_s_per_wavefront": round(bins_per_frame / (kt["entropy"] * 1e-3), 0) if bins_per_frame else None,
                          "bits_per_s_per_wavefront": round(bytes_per_frame * 8 / (kt["entropy"] * 1e-3), 0),
                          "kernel_ms": round(kt["entropy"], 3),
                          "note": "k_entropy + k_dbprep of a pass, one slice per wavefront, every slice resident for the whole kernel (the I slice's latency bounds it): per-wavefront rates are "
                                  "the average slice's; bins: the checker's count for stream 0 in the cpu_baseline leg x streams (null without that leg)"},
        "end_to_end_fps": round(S * F / e2e_s, 2),
        "k_pack": k_pack,
        "pipelined_ingest_fps": round(pipelined_fps, 2),
        "pipelined_ingest_note": "prepare(k+1) (host parse + H2D, second staging set) overlapped with execute(k); rank 0's own rate",
        "host_prepare_ms": round(prepare_s * 1e3, 2),
        "hbm_bytes": {"decoder": hbm_bytes, "per_stream": int(hbm_bytes / S), "coef_pool_used": round(pool_used / pool_cap, 3), "note": "device memory of the %d-stream decoder (I/P streams: what only B pictures need is allocated on demand)" % S},
        "parity": parity,
        "stream_gen_s": round(gen_s, 1),
        "stream_gen_note": "slowest rank; %d distinct streams per rank on %d generator threads" % (nd, gen_threads),
    }
    out.update(extra)
    print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)

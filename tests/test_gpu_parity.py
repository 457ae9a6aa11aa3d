"""GPU parity tests proper (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
must reproduce the oracle (and the generator's reconstruction) bit for bit."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import BASE, FIELD_CABAC_MATRIX, FIELD_MATRIX, FULL_MATRIX, MATRIX, POC_MATRIX, PRODUCT_DECODES_B, pictures_of

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "stream_md5.json")


class _x_wgs:
    """H264MI_X_WGS for the decoders created inside: 0 = a picture never leaves one workgroup in K3 / K5 (the kernels made
    for launches that fill the chip), n = spread over up to n workgroups per launch (default 256)."""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.old = os.environ.get("H264MI_X_WGS")
        if self.n is not None:
            os.environ["H264MI_X_WGS"] = str(self.n)

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("H264MI_X_WGS", None)
        else:
            os.environ["H264MI_X_WGS"] = self.old


def _nslices(kw):
    """slices per picture of a parity-matrix case (with slice groups: `slices` per group)"""
    return max(1, kw.get("slices", 1)) * max(1, kw.get("slice_groups", 1))


def _decode_gpu(H, streams, w, h, frames, slices=1, crop=False, x_wgs=None, **cfg):
    with _x_wgs(x_wgs):
        return _decode_gpu_(H, streams, w, h, frames, slices, crop, **cfg)


def _decode_gpu_(H, streams, w, h, frames, slices=1, crop=False, **cfg):
    W, Hc = (w + 15) // 16 * 16, (h + 15) // 16 * 16
    dec = H.Decoder(max_streams=len(streams), max_width=W, max_height=Hc, max_frames_per_batch=frames, max_slices_per_frame=max(1, slices),
                    max_bitstream_bytes=sum(len(s) for s in streams) * 2 + (1 << 20), **cfg)
    info = dec.decode(streams)
    size = (w * h if crop else W * Hc) * 3 // 2
    out = [dec.read_frames(i, crop=crop, size=size) for i in range(len(streams))]
    info.pocs = [[dec.frame_info(i, f).pic_order_cnt for f in range(dec.frame_count(i))] for i in range(len(streams))]
    dec.close()
    return out, info


@pytest.mark.parametrize("name", sorted(FULL_MATRIX))
def test_gpu_matches_oracle_and_generator(name, H, sg, oracle_mod):
    kw = FULL_MATRIX[name]
    stream, rec, _ = sg.encode(**kw)
    ref, _ = oracle_mod.decode(stream, crop=False)
    if kw.get("bframes") and not PRODUCT_DECODES_B:
        with pytest.raises(H.H264MIError) as e:
            _decode_gpu(H, [stream], kw["width"], kw["height"], pictures_of(kw), _nslices(kw))
        assert e.value.code == -3
        return
    out, info = _decode_gpu(H, [stream], kw["width"], kw["height"], pictures_of(kw), _nslices(kw))
    assert info.n_frames == kw["frames"]
    assert out[0].shape == ref.shape
    assert np.array_equal(out[0], ref), "GPU != oracle"
    assert np.array_equal(out[0], rec), "GPU != generator reconstruction"
    # picture order counts (8.2.1, incl. type 1 / type 2, non-reference pictures and the reset after MMCO 5)
    assert info.pocs[0] == list(oracle_mod.last_pocs) == list(sg.last_pocs()), "PicOrderCnt"


@pytest.mark.parametrize("name", sorted(POC_MATRIX))
def test_gpu_separate_bottom_field_counts(name, H, sg, oracle_mod):
    """bottom_field_pic_order_in_frame_present_flag = 1 (delta_pic_order_cnt_bottom / delta_pic_order_cnt[1], a bottom field that comes
    first or later, with B pictures, marking scripts and operation 5): pixels and PicOrderCnt as for the matrix.  (Run on the GPU at the
    end of round 3 through tools/poc_gpu_check.py: bit-exact, equal counts.)"""
    kw = POC_MATRIX[name]
    stream, rec, _ = sg.encode(**kw)
    dec = H.Decoder(max_streams=1, max_width=(kw["width"] + 15) & ~15, max_height=(kw["height"] + 15) & ~15, max_frames_per_batch=kw["frames"], max_slices_per_frame=8)
    try:
        dec.decode([stream])
        out = dec.read_frames(0, crop=False)
        assert out.shape == rec.shape and np.array_equal(out, rec), "GPU != generator reconstruction"
        assert [dec.frame_info(0, i).pic_order_cnt for i in range(kw["frames"])] == [int(x) for x in sg.last_pocs()], "PicOrderCnt"
    finally:
        dec.close()
    ref, _ = oracle_mod.decode(stream, crop=False)
    assert np.array_equal(out, ref), "GPU != oracle"


def test_gpu_matrix_with_one_workgroup_per_picture(H, sg):
    """The same matrix through k_intra / k_deblock / k_deblock_b (a picture inside ONE workgroup: what a launch that fills the
    chip uses); the default for these one-stream batches is the banded kernels.  Also 512 workgroups per launch."""
    for name in sorted(FULL_MATRIX):
        kw = FULL_MATRIX[name]
        stream, rec, _ = sg.encode(**kw)
        for x in (0, 512):
            out, _ = _decode_gpu(H, [stream], kw["width"], kw["height"], pictures_of(kw), _nslices(kw), x_wgs=x)
            assert np.array_equal(out[0], rec), (name, x)


@pytest.mark.parametrize("w,h", [(64, 2176), (48, 1120), (32, 4096), (2048, 48)])
def test_gpu_tall_and_wide_pictures_inside_one_workgroup(w, h, H, sg):
    """k_deblock hands rows from one group of eight macroblock rows to the next through LDS rings; a picture of more than 64 rows takes its eight wavefronts
    two rounds (1120 rows: 70 macroblock rows, 9 groups), more than 128 rows three and more -- there the ring of the last wavefront has a second buffer
    (2176 rows: 17 groups; 4096 rows: 32 groups, four rounds).  K3 walks the same pictures with one wavefront per macroblock row.  A wide, flat picture for
    the other extreme (128 macroblock columns, whole-row ring).  With and without B pictures' second list; both kernel families."""
    for kw in (dict(profile_idc=77, cabac=1, frames=3), dict(profile_idc=100, cabac=0, transform8x8=1, frames=4, bframes=1, num_ref_frames=2)):
        kw = dict(kw, width=w, height=h, idr_period=0, seed=77 + w)
        stream, rec, _ = sg.encode(**kw)
        for x in (0, 256):
            out, _ = _decode_gpu(H, [stream, stream], w, h, kw["frames"], x_wgs=x)
            assert np.array_equal(out[0], rec) and np.array_equal(out[1], rec), (w, h, kw["profile_idc"], x)


def test_gpu_golden_md5(H, sg):
    gold = json.load(open(GOLDEN))
    gold.update(json.load(open(os.path.join(os.path.dirname(GOLDEN), "field_md5.json"))))
    assert set(gold) == set(FULL_MATRIX)
    for name, g in sorted(gold.items()):
        kw = FULL_MATRIX[name]
        if kw.get("bframes") and not PRODUCT_DECODES_B:
            continue
        stream, _, _ = sg.encode(**kw)
        out, _ = _decode_gpu(H, [stream], kw["width"], kw["height"], pictures_of(kw), _nslices(kw))
        assert hashlib.md5(out[0].tobytes()).hexdigest() == g["frames_md5"], name


def test_gpu_batch_of_unequal_streams(H, sg, oracle_mod):
    """Several independent streams of different size, profile, entropy mode and length in ONE batch."""
    cfgs = [dict(width=176, height=144, frames=5, idr_period=0, profile_idc=77, cabac=1, seed=11),
            dict(width=64, height=48, frames=2, idr_period=1, profile_idc=66, cabac=0, seed=12),
            dict(width=180, height=100, frames=4, idr_period=0, profile_idc=100, cabac=1, transform8x8=1, slices=2, seed=13),
            dict(width=176, height=144, frames=1, idr_period=1, profile_idc=66, cabac=0, seed=14)]
    streams = [sg.encode(**c)[0] for c in cfgs]
    dec = H.Decoder(max_streams=5, max_width=192, max_height=144, max_frames_per_batch=5, max_slices_per_frame=2)
    dec.decode(streams + [b""])  # the last stream is empty
    for i, c in enumerate(cfgs):
        ref, info = oracle_mod.decode(streams[i], crop=True)
        assert dec.frame_count(i) == c["frames"]
        got = dec.read_frames(i, crop=True, size=info.width * info.height * 3 // 2)
        assert np.array_equal(got, ref), i
    assert dec.frame_count(4) == 0
    # K6 over the whole batch (odd sizes: 180x100 takes the byte path, 176x144 the 16-byte path)
    import torch
    want = np.concatenate([oracle_mod.decode(streams[i], crop=True)[0].reshape(-1) for i in range(4)])
    buf = torch.empty(want.size, dtype=torch.uint8, device="cuda")
    assert dec.pack_batch(buf.data_ptr(), buf.numel()) == want.size
    dec.sync()
    assert np.array_equal(buf.cpu().numpy(), want)
    dec.close()


def test_gpu_gop_split_across_calls(H, sg, oracle_mod):
    """Reference pictures persist between batches: a stream fed in two chunks decodes like one piece."""
    kw = dict(width=176, height=144, frames=6, idr_period=0, profile_idc=77, cabac=1, num_ref_frames=2, seed=21)
    stream, rec, sizes = sg.encode(**kw)
    cut = int(sizes[:3].sum())
    dec = H.Decoder(max_streams=1, max_width=176, max_height=144, max_frames_per_batch=3)
    dec.decode([stream[:cut]])
    a = dec.read_frames(0, crop=False)
    dec.decode([stream[cut:]])
    b = dec.read_frames(0, crop=False)
    dec.close()
    assert np.array_equal(np.concatenate([a, b]), rec)


def test_gpu_repeated_execute_is_idempotent(H, sg):
    kw = MATRIX["cabac_IPP"]
    stream, rec, _ = sg.encode(**kw)
    dec = H.Decoder(max_streams=1, max_width=176, max_height=144, max_frames_per_batch=kw["frames"])
    dec.prepare([stream])
    for _ in range(4):  # exercises the 3 buffer sets of the pass pipeline
        dec.execute()
    dec.sync()
    assert np.array_equal(dec.read_frames(0, crop=False), rec)
    dec.close()


@pytest.mark.parametrize("extra", [dict(frames=3), dict(frames=7, bframes=2, num_ref_frames=2, direct_temporal=1, weighted_bipred=2),
                                   dict(frames=7, bframes=2, num_ref_frames=3, cabac=0, sub8x8_permille=400),
                                   dict(frames=4, profile_idc=66, cabac=0, slice_groups=4, fmo_type=6, slices=2, aso=1),
                                   dict(frames=5, profile_idc=66, cabac=0, slice_groups=2, fmo_type=3, slices=2, aso=1)])
def test_gpu_corrupt_streams_do_not_hang_or_crash(extra, H, sg):
    kw = dict(dict(width=176, height=144, frames=3, idr_period=0, profile_idc=77, cabac=1, seed=5), **extra)
    stream, _, _ = sg.encode(**kw)
    rng = np.random.default_rng(0)
    for trial in range(6):
        b = bytearray(stream)
        for _ in range(8):
            i = int(rng.integers(60, len(b)))
            b[i] ^= 1 << int(rng.integers(0, 8))
        dec = H.Decoder(max_streams=1, max_width=176, max_height=144, max_frames_per_batch=kw["frames"])
        try:
            dec.decode([bytes(b)])
        except H.H264MIError as e:
            assert e.code in (-2, -3, -7, -8), e
        dec.close()
    # truncated
    dec = H.Decoder(max_streams=1, max_width=176, max_height=144, max_frames_per_batch=kw["frames"])
    try:
        dec.decode([stream[:len(stream) // 2]])
    except H.H264MIError as e:
        assert e.code in (-2, -8)
    dec.close()


def test_gpu_real_stream_matches_oracle(H, real_stream, oracle_mod):
    ref, info = oracle_mod.decode(real_stream, crop=True)
    out, _ = _decode_gpu(H, [real_stream], 320, 240, 36, crop=True)
    assert np.array_equal(out[0], ref)


def test_gpu_base_layer_of_streams_with_extension_nal_units(H, sg):
    """SVC / MVC / 3D-AVC streams (Annex G / H / J) carry their enhancement in NAL units of types 14, 15, 20, 21 around the base layer's own; the decoder
    decodes the base layer / base view and passes over the rest (h264/nalUnit.go:39-71 parses their header extension, h264/server.go:147-164 dispatches on
    types 1, 5, 7, 8 only)."""
    from conftest import with_extension_nals
    for kw in (dict(width=176, height=144, frames=6, idr_period=3, profile_idc=77, cabac=1, slices=2, seed=41),
               dict(width=176, height=144, frames=7, idr_period=0, profile_idc=100, cabac=0, transform8x8=1, bframes=2, num_ref_frames=3, seed=42),
               dict(width=320, height=192, frames=5, idr_period=0, profile_idc=66, cabac=0, slice_groups=3, fmo_type=1, slices=3, seed=43)):
        stream, rec, _ = sg.encode(**kw)
        ext = with_extension_nals(stream, seed=kw["seed"])
        out, info = _decode_gpu(H, [ext, stream], kw["width"], kw["height"], kw["frames"], _nslices(kw))
        assert info.n_frames == 2 * kw["frames"]
        assert np.array_equal(out[0], rec) and np.array_equal(out[1], rec), kw["seed"]


def test_gpu_1080p_full_size_properties(H, sg, oracle_mod):
    """BASELINE config at full size: 1080p Main CABAC IPPP, cropping 1088 -> 1080, two streams."""
    kw = sg.recipe("C3", frames=4, idr_period=4)
    streams = [sg.encode(**dict(kw, seed=3 + i))[0] for i in range(2)]
    out, info = _decode_gpu(H, streams, 1920, 1080, 4, crop=True)
    assert (info.width, info.height, info.coded_width, info.coded_height) == (1920, 1080, 1920, 1088)
    refs = [oracle_mod.decode(streams[i], crop=True)[0] for i in range(2)]
    for i in range(2):
        assert np.array_equal(out[i], refs[i])
    for x in (0, 7, 64):  # one workgroup per picture; 3 bands of 6 row groups; 32 bands
        out2, _ = _decode_gpu(H, streams, 1920, 1080, 4, crop=True, x_wgs=x)
        for i in range(2):
            assert np.array_equal(out2[i], refs[i]), x
    # device-side crop/pack kernel (K6) == host-side cropped read
    import torch
    dec = H.Decoder(max_streams=1, max_width=1920, max_height=1088, max_frames_per_batch=4)
    dec.decode([streams[0]])
    buf = torch.empty(1920 * 1080 * 3 // 2, dtype=torch.uint8, device="cuda")
    from h264decode_amd._lib import check
    check(dec._L.h264mi_frame_pack_device(dec._h, 0, 3, buf.data_ptr(), buf.numel()))
    dec.sync()
    assert np.array_equal(buf.cpu().numpy(), out[0][3])
    # ... and the batched form: all 4 frames of the stream in one launch
    allbuf = torch.empty(4 * 1920 * 1080 * 3 // 2, dtype=torch.uint8, device="cuda")
    assert dec.pack_batch(allbuf.data_ptr(), allbuf.numel()) == allbuf.numel()
    dec.sync()
    assert np.array_equal(allbuf.cpu().numpy().reshape(4, -1), out[0])
    dec.close()


def test_gpu_720p_cavlc_intra(H, sg, oracle_mod):
    """BASELINE configs[1]: 720p Baseline CAVLC I-frames."""
    kw = sg.recipe("C2", frames=2)
    stream, rec, _ = sg.encode(**kw)
    ref, _ = oracle_mod.decode(stream, crop=False)
    assert np.array_equal(ref, rec), "oracle != generator"
    for x in (None, 0):
        out, _ = _decode_gpu(H, [stream], 1280, 720, 2, x_wgs=x)
        assert np.array_equal(out[0], rec), x
        assert np.array_equal(out[0], ref), x


@pytest.mark.parametrize("kw", [dict(slice_groups=6, fmo_type=1, slices=2), dict(slice_groups=8, fmo_type=6, aso=1), dict(slice_groups=2, fmo_type=3, slices=3, aso=1),
                                dict(slice_groups=5, fmo_type=2, aso=1, interlace_sps=1, height=704)])
def test_gpu_720p_slice_groups(kw, H, sg, oracle_mod):
    """Slice groups at a size where a row is more than one 64-macroblock chunk (80 columns) and a slice visits thousands of scattered
    macroblocks: GPU == oracle == generator, one workgroup per picture and banded."""
    kw = dict(dict(width=1280, height=720, frames=3, idr_period=0, profile_idc=66, cabac=0, intra_in_p_permille=100, seed=31), **kw)
    stream, rec, _ = sg.encode(**kw)
    ref, _ = oracle_mod.decode(stream, crop=False)
    assert np.array_equal(ref, rec)
    for x in (None, 0):
        out, _ = _decode_gpu(H, [stream], 1280, kw["height"], 3, slices=_nslices(kw), x_wgs=x)
        assert np.array_equal(out[0], rec), x


def test_gpu_4k_high_8_slices(H, sg, oracle_mod):
    """BASELINE configs[3]: 3840x2160 High CABAC, 8x8 transform, 8 slices per picture (240 x 135 macroblocks: widest row
    state, 34 deblocking row groups in 3 rounds, slice boundaries inside and across macroblock rows)."""
    kw = sg.recipe("C4", frames=3, idr_period=3)
    stream, rec, sizes = sg.encode(**kw)
    # the oracle on the first two pictures (I, P) of the same stream: the full-size cases are held against BOTH independent implementations
    ref2, _ = oracle_mod.decode(stream[:int(sizes[:2].sum())], crop=False)
    assert np.array_equal(ref2, rec[:2]), "oracle != generator"
    for x in (None, 0, 3):  # 34 bands; one workgroup (3 rounds of 12 wavefronts); 3 bands of 12 wavefronts
        out, info = _decode_gpu(H, [stream], 3840, 2160, 3, slices=8, x_wgs=x)
        assert (info.coded_width, info.coded_height) == (3840, 2160)
        assert np.array_equal(out[0], rec), x


def test_gpu_sizing_knobs(H, sg):
    """h264mi_config.max_ref_frames / coef_blocks_per_mb: a smaller frame pool and residual pool decode what fits, bit-exactly, and refuse
    what does not with H264MI_ECAPACITY / a reported pool exhaustion -- never with a fault."""
    kw = dict(width=176, height=144, frames=6, idr_period=0, profile_idc=77, cabac=1, num_ref_frames=2, seed=51)
    stream, rec, _ = sg.encode(**kw)
    big = H.Decoder(max_streams=1, max_width=176, max_height=144, max_frames_per_batch=6)
    small = H.Decoder(max_streams=1, max_width=176, max_height=144, max_frames_per_batch=6, max_ref_frames=2)
    assert small.device_bytes() < big.device_bytes()
    assert big.device_bytes() - small.device_bytes() >= 14 * 176 * 144 * 3 // 2
    small.decode([stream])
    assert np.array_equal(small.read_frames(0, crop=False), rec)
    used, cap = small.coef_pool()
    assert 0 < used <= cap
    big.close()
    small.close()
    tight = H.Decoder(max_streams=1, max_width=176, max_height=144, max_frames_per_batch=6, max_ref_frames=1)
    with pytest.raises(H.H264MIError) as e:
        tight.decode([stream])  # the SPS declares two reference frames
    assert e.value.code == -7
    tight.close()


def test_gpu_residual_pool_exhaustion_is_recovered(H, sg):
    """The residual pool holds a configured number of 32-byte blocks per macroblock (default 8; the worst case is 26).  A batch that needs more than a
    pass's pool is repeated by the library on its own with the pools of all three record sets as one (h264mi_batch_sync: retry_exhausted) -- bit-exact,
    no stream marked -- and only what does not fit into that either fails (H264MI_EDECODE, entropy status 40), without a fault.  Streams of one
    batch share the pool: the I B B P stream here sits next to a plain one, which must come through whatever happens to its neighbour."""
    kws = [dict(width=1280, height=720, frames=7, idr_period=0, profile_idc=77, cabac=1, bframes=2, num_ref_frames=3, qp=18, noise=25, seed=71),
           dict(width=1280, height=720, frames=7, idr_period=0, profile_idc=77, cabac=1, qp=30, seed=72)]
    enc = [sg.encode(**kw) for kw in kws]
    mbs, head = 2 * 80 * 45 * 7, 15 * 2048  # macroblock records of the decoder below; the pool's chunk head-room (slices + 1 chunks of 2048 blocks)
    roomy = H.Decoder(max_streams=2, max_width=1280, max_height=720, max_frames_per_batch=7, max_slices_per_frame=1)
    roomy.decode([e[0] for e in enc])
    used, _ = roomy.coef_pool()
    roomy.close()
    # a pool the first pass overruns, while three of them are enough: blocks per macroblock k with k mbs + head < 0.8 used and 3 (k mbs + head) > 1.25 used
    ks = [k for k in range(1, 27) if k * mbs + head < 0.8 * used and 3 * (k * mbs + head) > 1.25 * used]
    assert ks, (used, mbs)  # (the test needs a batch with a real appetite)
    k = ks[0]
    dec = H.Decoder(max_streams=2, max_width=1280, max_height=720, max_frames_per_batch=7, max_slices_per_frame=1, coef_blocks_per_mb=k)
    dec.decode([e[0] for e in enc])
    used2, cap2 = dec.coef_pool()
    assert used2 > cap2, (used2, cap2)  # more than one set's pool was taken: the pass was repeated with all of it
    for i in range(2):
        assert dec.stream_status(i) == 0
        assert np.array_equal(dec.read_frames(i, crop=False), enc[i][1]), i
    # ... and again (the repeat leaves the decoder in order): the same batch, then it executed twice in a row
    dec.decode([e[0] for e in enc])
    dec.execute()
    dec.execute()
    dec.sync()
    assert np.array_equal(dec.read_frames(0, crop=False), enc[0][1])
    dec.close()
    # a pool so small that even the whole allocation falls short: a reported failure, not a fault
    if 3 * (mbs + head) < 0.8 * used:
        tiny = H.Decoder(max_streams=2, max_width=1280, max_height=720, max_frames_per_batch=7, max_slices_per_frame=1, coef_blocks_per_mb=1)
        with pytest.raises(H.H264MIError) as e:
            tiny.decode([e2[0] for e2 in enc])
        assert e.value.code == -8 and "40" in str(e.value)
        tiny.close()


def test_gpu_residual_pool_exhaustion_of_an_older_batch_is_reported_not_repeated(H, sg):
    """execute(n); prepare(n + 1); execute(n + 1); ONE sync, batch n + 1 continuing the GOP with one reference frame, and batch n overrunning its
    residual pool: batch n + 1 has by then written into the frame slots batch n's first P pictures predict from, so repeating batch n would
    predict from the wrong samples and report success (advisor, round 4).  The library repeats only the batch executed LAST; here the
    exhaustion must surface as a failure of the stream -- never as silently wrong pictures."""
    kw = dict(width=1280, height=720, frames=8, idr_period=0, profile_idc=77, cabac=1, num_ref_frames=1, qp=18, noise=25, seed=73)
    stream, rec, _ = sg.encode(**kw)
    nals = H.read_nal_units(stream)
    offs = [n.Offset - 4 for n in nals] + [len(stream)]
    pic = [i for i, n in enumerate(nals) if n.Type in (1, 5)]
    halves = [stream[:offs[pic[4]]], stream[offs[pic[4]]:]]
    mbs, head = 80 * 45 * 4, 3 * 2048
    roomy = H.Decoder(max_streams=1, max_width=1280, max_height=720, max_frames_per_batch=4, max_slices_per_frame=1)
    roomy.decode([halves[0]])
    used, _ = roomy.coef_pool()
    roomy.decode([halves[1]])
    assert np.array_equal(roomy.read_frames(0, crop=False), rec[4:])
    roomy.close()
    ks = [k for k in range(1, 27) if k * mbs + head < 0.8 * used and 3 * (k * mbs + head) > 1.25 * used]
    assert ks, (used, mbs)
    dec = H.Decoder(max_streams=1, max_width=1280, max_height=720, max_frames_per_batch=4, max_slices_per_frame=1, coef_blocks_per_mb=ks[0])
    dec.set_isolation(True)
    dec.prepare([halves[0]])
    dec.execute()
    dec.prepare([halves[1]])
    dec.execute()
    dec.sync()
    assert dec.stream_status(0) == -8, dec.stream_status(0)  # the first batch's exhaustion is held against the stream
    dec.close()
    # the same calls with a sync per batch: the first batch is the last one executed when it is looked at, is repeated with the whole pool, and all is well
    dec = H.Decoder(max_streams=1, max_width=1280, max_height=720, max_frames_per_batch=4, max_slices_per_frame=1, coef_blocks_per_mb=ks[0])
    dec.decode([halves[0]])
    assert dec.stream_status(0) == 0 and np.array_equal(dec.read_frames(0, crop=False), rec[:4])
    dec.decode([halves[1]])
    assert dec.stream_status(0) == 0 and np.array_equal(dec.read_frames(0, crop=False), rec[4:])
    dec.close()


def test_gpu_rejects_out_of_scope_profile(H):
    """A third-party High 4:4:4 Predictive stream (chroma_format_idc 3) must be refused with a clear status, not mis-decoded."""
    import os
    from mp4util import mp4_to_annexb
    path = "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/cockatoo.mp4"
    if not os.path.exists(path):
        pytest.skip("third-party sample MP4 not present on this machine")
    stream = mp4_to_annexb(open(path, "rb").read())
    dec = H.Decoder(max_streams=1, max_width=1280, max_height=720, max_frames_per_batch=8, max_slices_per_frame=4)
    with pytest.raises(H.H264MIError) as ei:
        dec.decode([stream[:400000]])
    assert ei.value.code == -3 and "4:2:0" in str(ei.value)  # H264MI_EUNSUPPORTED
    dec.close()


def test_gpu_data_partitioning_is_refused_with_a_reason(H, sg):
    """Slice data partitions (NAL unit types 2, 3, 4; Extended profile) are not decoded: a stream that carries one is refused with a status and a message --
    passing them over like SEI would silently drop its pictures."""
    stream, _, _ = sg.encode(width=176, height=144, frames=3, idr_period=0, profile_idc=66, cabac=0, seed=9)
    part = stream + b"\x00\x00\x01\x22" + b"\x9a\x55\xaa\x80"  # nal_ref_idc 1, nal_unit_type 2
    dec = H.Decoder(max_streams=2, max_width=176, max_height=144, max_frames_per_batch=4)
    try:
        with pytest.raises(H.H264MIError) as ei:
            dec.decode([part])
        assert ei.value.code == -3 and "partition" in str(ei.value)
    finally:
        dec.close()


def test_gpu_c_program_through_the_abi(H, sg, oracle_mod, tmp_path):
    """examples/h264mi_decode.c (plain C, only include/h264mi.h): file in, raw I420 out, several batches."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")])
    for kw in (dict(width=180, height=100, frames=8, idr_period=3, profile_idc=100, cabac=1, transform8x8=1, slices=2, long_start_code=0, seed=77),
               # slice groups in shuffled slice order: the program must not cut pictures at "first_mb_in_slice == 0"
               dict(width=176, height=144, frames=7, idr_period=4, profile_idc=66, cabac=0, slice_groups=3, fmo_type=6, slices=2, aso=1, seed=78)):
        stream = sg.encode(**kw)[0]
        ref, info = oracle_mod.decode(stream, crop=True)
        src, dst = tmp_path / "in.h264", tmp_path / "out.yuv"
        src.write_bytes(stream)
        subprocess.check_call([os.path.join(root, "examples", "h264mi_decode"), str(src), str(dst), "3"])
        got = np.frombuffer(dst.read_bytes(), dtype=np.uint8).reshape(-1, info.width * info.height * 3 // 2)
        assert np.array_equal(got, ref)



def test_gpu_c5_share_32_distinct_1080p_streams(H, sg, oracle_mod):
    """BASELINE configs[4], one GPU's share: 32 DISTINCT 1080p Main CABAC streams (seeds 1000..1031), one
    IPPP GOP of 30 frames each, decoded as one batch; every frame of every stream must equal the generator's own
    reconstruction (an implementation independent of the product and of the oracle)."""
    from concurrent.futures import ThreadPoolExecutor
    S, F = 32, 30
    kws = [sg.recipe("C3", frames=F, idr_period=F, seed=1000 + i) for i in range(S)]
    with ThreadPoolExecutor(max_workers=min(S, max(1, (os.cpu_count() or 8) - 2))) as ex:  # ctypes releases the GIL
        gen = list(ex.map(lambda kw: sg.encode(**kw), kws))
    streams = [g[0] for g in gen]
    assert len(set(hashlib.md5(s).hexdigest() for s in streams)) == S, "streams are not distinct"
    # the oracle on the first two pictures (IDR + P) of four of the streams: generator and oracle agree at this size too
    for si in (0, 7, 19, 31):
        ref2, _ = oracle_mod.decode(streams[si][:int(gen[si][2][:2].sum())], crop=False)
        assert np.array_equal(ref2, gen[si][1][:2]), "oracle != generator (stream %d)" % si
    for x in (None, 0):  # 8 bands per picture (the default for 32 pictures per launch), then one workgroup per picture
        with _x_wgs(x):
            dec = H.Decoder(max_streams=S, max_width=1920, max_height=1088, max_frames_per_batch=F, max_slices_per_frame=1,
                            max_bitstream_bytes=int(sum(len(s) for s in streams) * 1.1) + (1 << 20))
        dec.prepare(streams)
        for _ in range(3 if x is None else 1):  # pipelined passes: the entropy kernels of the next pass run beside the banded kernels (uneven load)
            dec.execute()
        dec.sync()
        fsz = 1920 * 1088 * 3 // 2
        for si in range(S):
            assert dec.frame_count(si) == F
            for f in range(F):
                got = dec.read_frame(si, f, crop=False)[:fsz]
                assert np.array_equal(got, gen[si][1][f]), "stream %d frame %d differs from the generator's reconstruction (x_wgs %s)" % (si, f, x)
        dec.close()


def _poison(H, dec):
    import ctypes
    f = H.load_hooks().h264mi_internal_poison
    f.restype, f.argtypes = ctypes.c_int32, [ctypes.c_void_p]
    assert f(dec._h) == 0


def test_gpu_lost_and_broken_slices_leave_defined_pictures(H, sg, oracle_mod):
    """Macroblock records no slice delivered must not be whatever an earlier batch (or the allocator) left in memory: every
    intermediate buffer is filled with 0xFF first, then a stream with a dropped slice NAL, one with a slice cut short, and
    corrupted ones are decoded -- in a decoder that held a differently sized batch before.  The undamaged slices must
    still decode exactly, lost macroblocks come out mid-grey, and two runs give identical pictures."""
    kw = dict(width=176, height=144, frames=2, idr_period=1, profile_idc=77, cabac=1, slices=3, seed=9, deblock_idc=2)
    stream, rec, _ = sg.encode(**kw)
    nals = H.read_nal_units(stream)
    # split at the NALs (the generator writes 4-byte start codes) to rebuild the stream without the 2nd slice of picture 0
    offs = [n.Offset - 4 for n in nals] + [len(stream)]

    def piece(i):
        return stream[offs[i]:offs[i + 1]]
    pieces = [piece(i) for i in range(len(nals))]
    assert b"".join(pieces) == stream
    assert [n.Type for n in nals] == [7, 8, 5, 5, 5, 7, 8, 5, 5, 5]
    lost = b"".join(p for i, p in enumerate(pieces) if i != 3)
    cut = b"".join(p[:len(p) // 2] if i == 4 else p for i, p in enumerate(pieces))
    outs = []
    for rep_ in range(2):
        dec = H.Decoder(max_streams=2, max_width=352, max_height=288, max_frames_per_batch=4, max_slices_per_frame=4)
        big = sg.encode(width=352, height=288, frames=3, idr_period=0, profile_idc=77, cabac=1, seed=3)[0]
        dec.decode([big, big])  # leaves records of another geometry behind
        _poison(H, dec)
        dec.set_isolation(True)
        dec.reset()
        dec.decode([lost, cut])
        assert dec.stream_status(0) == 0           # a missing slice is not an error of the slices that are there
        assert dec.stream_status(1) in (0, -8)     # the truncated slice may or may not trip the entropy decoder
        a = dec.read_frames(0, crop=False)
        b = dec.read_frames(1, crop=False)
        outs.append((a.copy(), b.copy()))
        dec.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), "pictures differ between runs"
    W, Hc = 176, 144
    y0 = outs[0][0][0][:W * Hc].reshape(Hc, W)
    ry = rec[0][:W * Hc].reshape(Hc, W)
    # 9 macroblock rows in 3 slices of 3 rows; deblocking does not cross slice edges (idc 2): rows of slices 0 and 2 are exact
    assert np.array_equal(y0[:48], ry[:48]) and np.array_equal(y0[96:], ry[96:])
    assert (y0[48:96] == 128).all(), "lost macroblocks are not mid-grey"
    assert np.array_equal(outs[0][0][1], rec[1])   # the next IDR picture is complete again
    yb = outs[0][1][0][:W * Hc].reshape(Hc, W)
    assert np.array_equal(yb[:48], ry[:48])        # stream 1: slice 0 intact, the cut slice 1 ends early somewhere


def test_gpu_slice_group_streams_share_a_batch_with_plain_ones(H, sg):
    """A launch that holds ONE picture with slice groups runs the slice-group build of the entropy kernel for all its slices: plain
    CABAC / CAVLC streams, a multi-slice one and two slice-group streams (explicit map with arbitrary slice order; evolving box-out) side
    by side, in one workgroup per picture and banded."""
    kws = [dict(width=176, height=144, frames=5, idr_period=0, profile_idc=77, cabac=1, seed=21),
           dict(width=176, height=144, frames=5, idr_period=0, profile_idc=66, cabac=0, slice_groups=4, fmo_type=6, slices=2, aso=1, seed=22),
           dict(width=176, height=144, frames=5, idr_period=0, profile_idc=100, cabac=1, transform8x8=1, slices=3, seed=23),
           dict(width=176, height=144, frames=5, idr_period=0, profile_idc=66, cabac=0, slice_groups=2, fmo_type=3, aso=1, seed=24),
           dict(width=176, height=144, frames=5, idr_period=0, profile_idc=66, cabac=0, seed=25)]
    gen = [sg.encode(**kw) for kw in kws]
    for x in (None, 0):
        out, info = _decode_gpu(H, [g[0] for g in gen], 176, 144, 5, slices=8, x_wgs=x)
        assert info.n_frames == 25
        for i, g in enumerate(gen):
            assert np.array_equal(out[i], g[1]), (i, x)


def test_gpu_lost_slice_of_a_slice_group(H, sg):
    """Slice groups scatter a slice over the picture, so its wavefront cannot blank a contiguous range for what it does not deliver:
    the records of such a picture are zeroed before the entropy kernels run.  One of three slices (groups) of a dispersed map is
    dropped, the buffers are poisoned first: the other two groups decode exactly (slice-edge deblocking is off), the lost
    macroblocks -- every third one, staggered from row to row -- are mid-grey."""
    kw = dict(width=176, height=144, frames=2, idr_period=1, profile_idc=66, cabac=0, slice_groups=3, fmo_type=1, seed=11, deblock_idc=2)
    stream, rec, _ = sg.encode(**kw)
    nals = H.read_nal_units(stream)
    assert [n.Type for n in nals] == [7, 8, 5, 5, 5, 7, 8, 5, 5, 5]
    offs = [n.Offset - 4 for n in nals] + [len(stream)]
    lost = b"".join(stream[offs[i]:offs[i + 1]] for i in range(len(nals)) if i != 3)
    sps = H.NewSPS(nals[0].RBSP())
    pps = H.NewPPS(sps, nals[1].RBSP())
    m = H.MbToSliceGroupMap(sps, pps).reshape(9, 11)
    first = H.NewSliceContext(H.VideoStream(sps, pps), nals[3], nals[3].RBSP()).Slice.Header.FirstMbInSlice
    gone = m[first // 11][first % 11]
    dec = H.Decoder(max_streams=1, max_width=352, max_height=288, max_frames_per_batch=4, max_slices_per_frame=4)
    dec.decode([sg.encode(width=352, height=288, frames=3, idr_period=0, profile_idc=77, cabac=1, seed=3)[0]])
    _poison(H, dec)
    dec.reset()
    dec.decode([lost])
    out = dec.read_frames(0, crop=False)
    dec.close()
    y, ry = out[0][:176 * 144].reshape(144, 176), rec[0][:176 * 144].reshape(144, 176)
    n_gone = 0
    for my in range(9):
        for mx in range(11):
            blk, rblk = y[my * 16:my * 16 + 16, mx * 16:mx * 16 + 16], ry[my * 16:my * 16 + 16, mx * 16:mx * 16 + 16]
            if m[my][mx] == gone:
                assert (blk == 128).all(), (mx, my)
                n_gone += 1
            else:
                assert np.array_equal(blk, rblk), (mx, my)
    assert n_gone == int((m == gone).sum()) and n_gone > 25
    assert np.array_equal(out[1], rec[1])  # the next IDR picture is complete again


def test_gpu_isolation_keeps_other_streams_alive(H, sg, oracle_mod):
    """One malformed stream in a batch (P pictures whose IDR picture is missing) is dropped and marked; the others decode."""
    good = sg.encode(width=176, height=144, frames=3, idr_period=0, profile_idc=77, cabac=1, seed=41)
    other = sg.encode(width=176, height=144, frames=3, idr_period=0, profile_idc=77, cabac=1, seed=42)[0]
    offs = [n.Offset - 4 for n in H.read_nal_units(other)]
    bad = other[:offs[2]] + other[offs[3]:]  # SPS, PPS, then P pictures without their IDR picture: H264MI_EBITSTREAM
    dec = H.Decoder(max_streams=3, max_width=176, max_height=144, max_frames_per_batch=3)
    with pytest.raises(H.H264MIError):
        dec.decode([good[0], bad, good[0]])  # without isolation the batch fails as before
    dec.close()
    dec = H.Decoder(max_streams=3, max_width=176, max_height=144, max_frames_per_batch=3)
    dec.set_isolation(True)
    dec.decode([good[0], bad, good[0]])
    assert dec.stream_status(0) == 0 and dec.stream_status(2) == 0 and dec.stream_status(1) == -2
    assert dec.frame_count(1) == 0
    dec.decode([good[0], bad, good[0]])  # the marked stream now waits for an IDR picture: its P pictures are skipped, not errors
    assert dec.stream_status(1) == 0 and dec.frame_count(1) == 0 and dec.frame_count(0) == 3
    assert np.array_equal(dec.read_frames(0, crop=False), good[1]) and np.array_equal(dec.read_frames(2, crop=False), good[1])
    # the slot is given to a new client: nothing of the old one survives, a P picture without an IDR is not decodable
    dec.reset_stream(1)
    dec.decode([b"", good[0], b""])
    assert dec.stream_status(1) == 0 and np.array_equal(dec.read_frames(1, crop=False), good[1])
    dec.close()


def test_gpu_lost_reference_pictures_are_refused_not_mispredicted(H, sg, oracle_mod):
    """8.2.5.2 without gaps_in_frame_num_value_allowed_flag: a frame_num that skips values means reference pictures were lost.
    The stream is refused (H264MI_EBITSTREAM; with isolation it alone leaves the batch) instead of predicting from whatever
    happens to sit in the lists; the same stream with the flag set decodes (matrix cases fn_gaps_*).  Also the real thing:
    a reference picture NAL cut out of a healthy stream."""
    kw = dict(width=176, height=144, frames=8, idr_period=0, profile_idc=77, cabac=1, num_ref_frames=4, fn_gap_period=3, seed=74)
    lossy = sg.encode(**dict(kw, fn_gap_declared=0))[0]
    good = sg.encode(**dict(kw, fn_gap_declared=1))
    with pytest.raises(RuntimeError):
        oracle_mod.decode(lossy, crop=False)
    whole = sg.encode(width=176, height=144, frames=6, idr_period=0, profile_idc=66, cabac=0, num_ref_frames=2, seed=75)[0]
    offs = [n.Offset - 4 for n in H.read_nal_units(whole)] + [len(whole)]
    cut = whole[:offs[4]] + whole[offs[5]:]  # SPS PPS IDR P | P(frame_num 2) removed | P P ...
    for bad in (lossy, cut):
        dec = H.Decoder(max_streams=2, max_width=176, max_height=144, max_frames_per_batch=8)
        with pytest.raises(H.H264MIError) as ei:
            dec.decode([bad, good[0]])
        assert ei.value.code == -2 and "frame_num" in str(ei.value)
        dec.close()
        dec = H.Decoder(max_streams=2, max_width=176, max_height=144, max_frames_per_batch=8)
        dec.set_isolation(True)
        dec.decode([bad, good[0]])
        assert dec.stream_status(0) == -2 and dec.frame_count(0) == 0 and dec.stream_status(1) == 0
        assert np.array_equal(dec.read_frames(1, crop=False), good[1])
        dec.close()


def test_gpu_pipelined_batches_do_not_lose_a_failure(H, sg):
    """execute(k); prepare(k + 1); execute(k + 1); ... with ONE sync at the end: a slice that fails in the entropy kernel in
    batch 2 must still take its stream out (nothing decodable before the next IDR picture) -- its status words used to be
    looked at only for the batch executed last."""
    good = sg.encode(width=176, height=144, frames=12, idr_period=0, profile_idc=66, cabac=0, seed=91)
    other = sg.encode(width=176, height=144, frames=12, idr_period=0, profile_idc=66, cabac=0, seed=92)[0]

    def chunks(stream):
        nals = H.read_nal_units(stream)
        offs = [n.Offset - 4 for n in nals] + [len(stream)]
        pic = [i for i, n in enumerate(nals) if n.Type in (1, 5)]
        cuts = [0] + [offs[pic[k]] for k in (3, 6, 9)] + [len(stream)]
        return [stream[cuts[i]:cuts[i + 1]] for i in range(4)], nals, offs
    gch, _, _ = chunks(good[0])
    och, nals, offs = chunks(other)
    # the first P picture of chunk 2 (picture 3): slice header kept, slice data replaced by 00000001 bytes -> an mb_skip_run / mb_type of 127 within the first two syntax elements
    k = [i for i, n in enumerate(nals) if n.Type in (1, 5)][3]
    sps = H.NewSPS(nals[0].RBSP())
    hdr = H.NewSliceContext(H.VideoStream(sps, H.NewPPS(sps, nals[1].RBSP())), nals[k], nals[k].RBSP()).Slice.Header
    keep = 4 + 1 + (hdr.slice_data_bit_offset + 7) // 8  # start code, NAL header byte, every byte the slice header has bits in
    broken = other[offs[k]:offs[k] + keep] + b"\x01" * 24
    och[1] = broken + other[offs[k + 1]:offs[[i for i, n in enumerate(nals) if n.Type in (1, 5)][6]]]
    dec = H.Decoder(max_streams=2, max_width=176, max_height=144, max_frames_per_batch=3)
    dec.set_isolation(True)
    dec.prepare([gch[0], och[0]])
    dec.execute()
    dec.prepare([gch[1], och[1]])
    dec.execute()
    dec.prepare([gch[2], och[2]])  # takes the first batch's staging set back; batch 2's failure is not known yet
    dec.execute()
    dec.sync()                     # looks at batch 2 and batch 3
    assert dec.stream_status(1) == -8 and dec.stream_status(0) == 0
    assert np.array_equal(dec.read_frames(0, crop=False), good[1][6:9])
    dec.decode([gch[3], och[3]])   # P pictures only: the marked stream waits for an IDR picture
    assert dec.frame_count(1) == 0 and dec.frame_count(0) == 3 and np.array_equal(dec.read_frames(0, crop=False), good[1][9:12])
    dec.close()


def test_gpu_reset_discards_failures_of_abandoned_batches(H, sg):
    """Found by tools/api_fuzz.py: a batch that failed is executed again without a sync, then the caller resets the decoder (or the stream:
    a new connection takes the slot) and decodes something else -- the abandoned batch's failure must not be reported against the new
    stream, neither by h264mi_batch_sync nor in its status."""
    bad = sg.encode(width=96, height=80, frames=4, idr_period=2, profile_idc=66, cabac=0, slice_groups=3, fmo_type=1, aso=1, seed=2)[0]
    good, rec, _ = sg.encode(width=176, height=144, frames=5, idr_period=0, profile_idc=77, cabac=1, seed=1)
    for how in ("decoder", "stream"):
        dec = H.Decoder(max_streams=2, max_width=176, max_height=144, max_frames_per_batch=6, max_slices_per_frame=8)
        with pytest.raises(H.H264MIError):
            dec.decode([bad[:len(bad) * 2 // 3], b""])  # a slice cut short: entropy failure
        dec.execute()  # the same batch again, nobody synchronises on it
        if how == "decoder":
            dec.reset()
        else:
            dec.reset_stream(0)
        dec.decode([good, good])  # must not raise
        assert dec.stream_status(0) == 0 and dec.stream_status(1) == 0
        assert np.array_equal(dec.read_frames(0, crop=False), rec) and np.array_equal(dec.read_frames(1, crop=False), rec)
        dec.close()


def test_gpu_banded_kernels_across_the_epoch_wrap(H, sg):
    """The banded kernels tag their hand-off words with a 32-bit launch epoch and draw tickets from 32-bit counters: a decoder whose counters
    stand a few launches before the wrap (test hook h264mi_internal_set_epoch of the hooks build) must decode exactly -- tickets are compared
    modulo 2^32, the rings are zeroed when the epoch restarts at 1."""
    import ctypes
    kw = dict(width=176, height=144, frames=8, idr_period=0, profile_idc=77, cabac=1, num_ref_frames=2, intra_in_p_permille=150, seed=85)
    stream, rec, _ = sg.encode(**kw)
    f = H.load_hooks().h264mi_internal_set_epoch
    f.restype, f.argtypes = ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint32]
    for start in (0xFFFFFFF0, 0xFFFFFFFE, 0xFFFFFF00):
        dec = H.Decoder(max_streams=1, max_width=176, max_height=144, max_frames_per_batch=8)
        assert f(dec._h, start) == 0
        for _ in range(3):  # 3 x 8 pictures x 2 banded launches: crosses the wrap for the first two start values
            dec.decode([stream])
            assert np.array_equal(dec.read_frames(0, crop=False), rec), start
        dec.close()


def test_gpu_decoders_in_several_threads(H, sg):
    """include/h264mi.h, "Threading": a handle is used by one thread at a time, distinct handles are independent.  Four threads, each with
    its own decoder and its own kind of stream (CABAC, CAVLC with slice groups, B pictures, High 8x8 multi-slice), decode side by side for a
    few rounds; every thread's frames are its generator's, and an error raised in one thread carries that thread's message."""
    import threading
    kws = [dict(width=176, height=144, frames=6, idr_period=0, profile_idc=77, cabac=1, seed=81),
           dict(width=96, height=80, frames=5, idr_period=2, profile_idc=66, cabac=0, slice_groups=3, fmo_type=6, aso=1, seed=82),
           dict(width=176, height=144, frames=7, idr_period=0, profile_idc=77, cabac=1, bframes=2, num_ref_frames=3, seed=83),
           dict(width=180, height=100, frames=4, idr_period=0, profile_idc=100, cabac=1, transform8x8=1, slices=3, seed=84)]
    gen = [sg.encode(**kw) for kw in kws]
    results, errors = {}, {}

    def worker(i):
        try:
            kw, (stream, rec, _) = kws[i], gen[i]
            W, Hc = (kw["width"] + 15) // 16 * 16, (kw["height"] + 15) // 16 * 16
            dec = H.Decoder(max_streams=1, max_width=W, max_height=Hc, max_frames_per_batch=kw["frames"], max_slices_per_frame=8)
            ok = True
            for _ in range(6):
                dec.decode([stream])
                ok = ok and np.array_equal(dec.read_frames(0, crop=False), rec)
            try:
                dec.decode([stream[:len(stream) // 2 + i]])  # every thread ends with a failure of its own
            except H.H264MIError as e:
                errors[i] = str(e)
            dec.close()
            results[i] = ok
        except Exception as e:  # noqa: BLE001
            results[i] = repr(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert results == {0: True, 1: True, 2: True, 3: True}, results


def test_gpu_resolution_change_inside_one_batch(H, sg):
    """Two sequences of different size back to back in ONE chunk of one stream: every picture keeps its own geometry."""
    a = sg.encode(width=176, height=144, frames=3, idr_period=0, profile_idc=77, cabac=1, seed=51)
    b = sg.encode(width=320, height=200, frames=2, idr_period=0, profile_idc=66, cabac=0, seed=52)
    dec = H.Decoder(max_streams=1, max_width=320, max_height=208, max_frames_per_batch=5)
    dec.decode([a[0] + b[0]])
    assert dec.frame_count(0) == 5
    for f in range(3):
        fi = dec.frame_info(0, f)
        assert (fi.width, fi.height, fi.coded_width, fi.coded_height) == (176, 144, 176, 144)
        assert np.array_equal(dec.read_frame_tight(0, f, crop=False), a[1][f])
    for f in range(2):
        fi = dec.frame_info(0, 3 + f)
        assert (fi.width, fi.height, fi.coded_width, fi.coded_height) == (320, 200, 320, 208)
        assert np.array_equal(dec.read_frame_tight(0, 3 + f, crop=False), b[1][f])
    dec.close()


def test_gpu_repeated_execute_with_marking_inside_the_batch(H, sg):
    """h264mi_batch_execute may be repeated for a batch that does NOT start with an IDR picture and whose marking
    operations free reference pictures half-way: those slots must not be recycled inside the batch."""
    kw = dict(MATRIX["mmco_cabac"])
    stream, rec, sizes = sg.encode(**kw)
    cut = int(sizes[:7].sum())
    dec = H.Decoder(max_streams=1, max_width=176, max_height=144, max_frames_per_batch=kw["frames"])
    dec.decode([stream[:cut]])
    dec.prepare([stream[cut:]])
    for _ in range(3):
        dec.execute()
    dec.sync()
    assert np.array_equal(dec.read_frames(0, crop=False), rec[7:])
    dec.close()


def test_gpu_prepare_overlaps_execute(H, sg):
    """Two staging sets: h264mi_batch_prepare(n + 1) may run while batch n executes; sync and the frame accessors keep
    referring to the batch executed last, and its frames stay intact until the prepare after next."""
    a = sg.encode(width=176, height=144, frames=4, idr_period=0, profile_idc=77, cabac=1, seed=61)
    b = sg.encode(width=176, height=144, frames=3, idr_period=0, profile_idc=66, cabac=0, seed=62)
    c = sg.encode(width=176, height=144, frames=4, idr_period=0, profile_idc=100, cabac=1, transform8x8=1, seed=63)
    dec = H.Decoder(max_streams=1, max_width=176, max_height=144, max_frames_per_batch=4)
    dec.prepare([a[0]])
    dec.execute()            # batch A in flight
    dec.prepare([b[0]])      # host parse + H2D of batch B meanwhile
    dec.sync()
    assert dec.frame_count(0) == 4 and np.array_equal(dec.read_frames(0, crop=False), a[1])  # still batch A
    dec.execute()            # batch B
    dec.prepare([c[0]])      # reuses A's staging set; A's frames may go, B's must stay
    dec.sync()
    assert dec.frame_count(0) == 3 and np.array_equal(dec.read_frames(0, crop=False), b[1])
    dec.execute()
    dec.sync()
    assert np.array_equal(dec.read_frames(0, crop=False), c[1])
    dec.close()


# ---------------------------------------------------------------------------------------------------
# B pictures

@pytest.mark.parametrize("name,chunk", [("b_temporal_cabac", 1), ("b_ibbp_cavlc", 2), ("b_wp_implicit", 1), ("b_gop_intra_pcm", 3), ("b_pyramid_cabac", 1),
                                        ("b_pyramid_cavlc", 2), ("b_pyramid_implicit", 3)])
def test_gpu_b_stream_split_across_batches(name, chunk, H, sg):
    """Reference lists, picture order counts and the co-located motion (ColRec) survive batch boundaries: a B stream fed
    `chunk` pictures per call -- every B picture's co-located picture then comes from an earlier batch -- decodes like one piece."""
    kw = MATRIX[name]
    stream, rec, sizes = sg.encode(**kw)
    dec = H.Decoder(max_streams=1, max_width=kw["width"], max_height=kw["height"], max_frames_per_batch=chunk, max_slices_per_frame=_nslices(kw))
    got, pos = [], 0
    for k in range(0, len(sizes), chunk):
        n = int(sizes[k:k + chunk].sum())
        dec.decode([stream[pos:pos + n]])
        pos += n
        got.append(dec.read_frames(0, crop=False))
    dec.close()
    assert np.array_equal(np.concatenate(got), rec)


def test_gpu_b_and_p_streams_side_by_side_1080p(H, sg):
    """Full-size pictures through the two-list kernels (K4/K5 variants run next to the I/P ones in the same waves): four
    1080p streams -- temporal direct + implicit weights (CABAC), spatial direct (CAVLC), P only, High 8x8 with 4 slices."""
    base = dict(width=1920, height=1080, frames=7, idr_period=0, qp=30)
    cfgs = [dict(base, profile_idc=77, cabac=1, bframes=2, num_ref_frames=3, direct_temporal=1, weighted_bipred=2, bskip_permille=400, seed=61),
            dict(base, profile_idc=77, cabac=0, bframes=1, num_ref_frames=2, bskip_permille=400, seed=62),
            dict(base, profile_idc=77, cabac=1, num_ref_frames=2, seed=63),
            dict(base, profile_idc=100, cabac=1, transform8x8=1, bframes=3, num_ref_frames=2, slices=4, weighted_bipred=1, sub8x8_permille=300, seed=64)]
    enc = [sg.encode(**c) for c in cfgs]
    dec = H.Decoder(max_streams=4, max_width=1920, max_height=1088, max_frames_per_batch=7, max_slices_per_frame=4)
    dec.decode([e[0] for e in enc])
    for i, e in enumerate(enc):
        got = dec.read_frames(i, crop=False)
        assert got.shape == e[1].shape
        bad = [f for f in range(got.shape[0]) if not np.array_equal(got[f], e[1][f])]
        assert not bad, (i, bad)
    dec.close()


@pytest.mark.parametrize("name", ["b_ibbp_cabac", "b_gop_intra_pcm", "sub8x8_heavy", "b_pyramid_cavlc"])
def test_gpu_slice_data_carries_the_coded_mb_type(name, H, sg, oracle_mod):
    """NewSliceData (h264/slice.go:570): mb_type as coded -- every value of Tables 7-11 / 7-13 / 7-14, the intra types inside P
    and B slices with their offsets 5 / 23, P_Skip / B_Skip as inferred -- against what the oracle parsed, macroblock by
    macroblock; list-1 fields of B macroblocks; MbPred returns the mb_pred() fields."""
    kw = MATRIX[name]
    stream, _, _ = sg.encode(**kw)
    _, info, tr = oracle_mod.decode(stream, crop=False, trace=True)
    W, Hc = (kw["width"] + 15) // 16 * 16, (kw["height"] + 15) // 16 * 16
    nmb = (W // 16) * (Hc // 16)
    tr = tr.reshape(-1, nmb, 8)
    dec = H.Decoder(max_streams=1, max_width=W, max_height=Hc, max_frames_per_batch=kw["frames"], max_slices_per_frame=_nslices(kw))
    dec.decode([stream])
    nals = H.read_nal_units(stream)
    sps = H.NewSPS(nals[0].RBSP())
    vs = H.VideoStream(sps, H.NewPPS(sps, nals[1].RBSP()))
    seen, f = set(), -1
    for n in nals:
        if n.Type == 7:
            sps = H.NewSPS(n.RBSP())
        elif n.Type == 8:
            vs = H.VideoStream(sps, H.NewPPS(sps, n.RBSP()))
        if n.Type not in (1, 5):
            continue
        ctx = H.NewSliceContext(vs, n, n.RBSP())
        if ctx.Slice.Header.FirstMbInSlice != 0:
            continue  # (one call per picture; the cases here have one slice type per picture)
        f += 1
        st = ctx.Slice.Header.SliceType % 5
        sds = H.NewSliceData(ctx, decoder=dec, stream=0, frame=f)
        assert len(sds) == nmb
        for m, sd in enumerate(sds):
            raw = int(tr[f, m, 0])
            if raw == -1:
                assert sd.MbSkipFlag and sd.MbType == H.MB_TYPE_INFERRED and sd.MbTypeName in ("P_Skip", "B_Skip"), (f, m)
            else:
                assert not sd.MbSkipFlag and sd.MbType == raw, (f, m, sd.MbType, raw, sd.MbTypeName)
            seen.add((st, sd.MbTypeName))
            mp = H.MbPred(sd)
            if st == 1 and sd.RefIdxL0:
                assert len(sd.RefIdxL1) == 4 and len(sd.MvL1) == 16 and mp["MvL1"] == sd.MvL1
                assert all((r0 >= 0 or r1 >= 0) for r0, r1 in zip(sd.RefIdxL0, sd.RefIdxL1)), (f, m)  # every 8x8 block of an inter macroblock predicts from something
            if sd.SubMbType:
                assert sd.MbTypeName in ("P_8x8", "B_8x8") and all(0 <= t <= (12 if st == 1 else 3) for t in sd.SubMbType)
    assert f + 1 == kw["frames"]
    if kw.get("bframes"):
        names = {n for s_, n in seen if s_ == 1}
        assert len(names) >= 8, sorted(names)
        if kw.get("bskip_permille"):
            assert {"B_Skip", "B_Direct_16x16"} <= names, sorted(names)
        if name == "b_ibbp_cabac":
            assert "B_8x8" in names and len(names) >= 15, sorted(names)
        if name == "b_gop_intra_pcm":
            assert any(n.startswith("I_") for n in names), sorted(names)  # intra macroblocks inside B slices: mb_type 23 + the I-slice value
    dec.close()


def test_gpu_output_order_of_b_streams(H, sg):
    kw = MATRIX["b_gop_intra_pcm"]  # two IDR periods, three B pictures between the anchors
    stream, _, _ = sg.encode(**kw)
    dec = H.Decoder(max_streams=2, max_width=kw["width"], max_height=kw["height"], max_frames_per_batch=kw["frames"])
    pstream, _, _ = sg.encode(**MATRIX["cabac_IPP"])
    dec.decode([stream, pstream])
    order = dec.output_order(0)
    info = [dec.frame_info(0, f) for f in range(dec.frame_count(0))]
    assert sorted(order) == list(range(kw["frames"])) and order != list(range(kw["frames"]))
    seq, cur = [], -1
    for fi in info:
        cur += fi.idr
        seq.append(cur)
    keys = [(seq[i], info[i].pic_order_cnt) for i in order]
    assert keys == sorted(keys) and len(set(keys)) == len(keys)
    assert dec.output_order(1) == list(range(dec.frame_count(1)))  # no B pictures: decoding order
    dec.close()


# ------------------------------------------------------------------ field pictures (PAFF; SURVEY 8f rank 3, h264/slice.go:867-872, h264/sps.go:316-322)
def _access_units(H, stream):
    sp = H.AccessUnitSplitter(max_units_per_chunk=1)
    return sp.feed(stream) + sp.flush()




def test_gpu_reconstruction_waves_follow_the_prediction_dependencies(H, sg):
    """Pictures that do not predict from one another are reconstructed side by side (mi_api.cpp: Stage::pic_wave): an all-intra stream is ONE
    launch of K3 / K5 however many pictures it has, two GOPs in one batch share their waves, the B pictures between two anchors share one wave,
    and a batch that continues a GOP starts at wave 0 again -- with the pictures bit-exact in every case.  (The launch count is read through the
    profiling accessor: one K3 launch per wave.)"""
    def waves_and_parity(streams, recs, frames_per_batch, slices=1):
        W, Hc = (BASE["width"] + 15) & ~15, (BASE["height"] + 15) & ~15
        dec = H.Decoder(max_streams=len(streams), max_width=W, max_height=Hc, max_frames_per_batch=frames_per_batch, max_slices_per_frame=8)
        dec.set_profiling(True)
        dec.decode(streams)
        n = len(dec.launch_times_ms("intra"))
        for i, rec in enumerate(recs):
            assert np.array_equal(dec.read_frames(i, crop=False), rec), "stream %d" % i
        dec.close()
        return n
    base = dict(BASE, profile_idc=77, cabac=1)
    # all-intra: 6 IDR pictures, one wave
    s_i, r_i, _ = sg.encode(**dict(base, frames=6, idr_period=1, seed=901))
    assert waves_and_parity([s_i], [r_i], 6) == 1
    # I P P P P: five waves; two such GOPs in one batch: still five, the second I picture joins wave 0
    s_p, r_p, _ = sg.encode(**dict(base, frames=10, idr_period=5, seed=902))
    assert waves_and_parity([s_p], [r_p], 10) == 5
    # side by side with the all-intra stream: the longest chain decides
    assert waves_and_parity([s_p, s_i], [r_p, r_i], 10) == 5
    # I B B P B B P ...: an anchor per wave, the B pictures ride with the anchor that follows them in decoding order
    kb = dict(MATRIX["b_ibbp_cabac"])
    s_b, r_b, _ = sg.encode(**kb)
    anchors = 1 + (kb["frames"] - 1 + kb["bframes"]) // (kb["bframes"] + 1)
    dec_order_waves = waves_and_parity([s_b], [r_b], kb["frames"])
    assert dec_order_waves <= anchors + 1 and dec_order_waves < kb["frames"]

@pytest.mark.parametrize("name", ["field_IP", "field_bottom_first", "field_mixed_paff", "field_b_temporal", "field_mmco1_rplm_mixed", "field_fmo_boxout_mixed_aso"])
def test_gpu_field_stream_fed_picture_by_picture(name, H, sg):
    """One access unit (= one picture: a field is one) per batch: the two fields of a frame arrive in DIFFERENT batches.  The frame goes out
    once, with the batch of its second field; everything equals the generator's reconstruction, PicOrderCnt included."""
    kw = FIELD_MATRIX[name]
    stream, rec, _ = sg.encode(**kw)
    pocs = [int(x) for x in sg.last_pocs()]
    aus = _access_units(H, stream)
    assert len(aus) > kw["frames"]
    # (b_pictures: the co-located field of the stream's first B field lies two batches back here -- its motion must have been kept from the start)
    dec = H.Decoder(max_streams=1, max_width=(kw["width"] + 15) & ~15, max_height=(kw["height"] + 15) & ~15, max_frames_per_batch=2, max_slices_per_frame=max(8, _nslices(kw)),
                    b_pictures=1 if kw.get("bframes") else 0)
    frames, got_pocs, per_batch = [], [], []
    for au in aus:
        dec.decode([au])
        per_batch.append(dec.frame_count(0))
        for f in range(dec.frame_count(0)):
            frames.append(dec.read_frame_tight(0, f, crop=False))
            got_pocs.append(dec.frame_info(0, f).pic_order_cnt)
    dec.close()
    assert max(per_batch) == 1 and sum(per_batch) == kw["frames"] and 0 in per_batch  # first fields deliver nothing
    assert got_pocs == pocs
    assert np.array_equal(np.stack(frames), rec), "GPU != generator reconstruction"


def test_gpu_single_field_at_the_end(H, sg, oracle_mod):
    """A first field whose second field never comes: held back until something says that it will not come -- here an end-of-stream NAL unit --,
    then it goes out as a frame with its rows decoded and the other parity mid-grey (what the oracle makes of the cut stream at its end)."""
    kw = dict(FIELD_MATRIX["field_IP"], slices=1)
    stream, rec, _ = sg.encode(**kw)
    cut = stream.rfind(b"\x00\x00\x01")
    cut -= 1 if stream[cut - 1] == 0 else 0
    ref, info = oracle_mod.decode(stream[:cut], crop=False)
    assert info.n_frames == kw["frames"]
    W, Hc = (kw["width"] + 15) & ~15, (kw["height"] + 15) & ~15
    dec = H.Decoder(max_streams=1, max_width=W, max_height=Hc, max_frames_per_batch=2 * kw["frames"], max_slices_per_frame=8)
    dec.decode([stream[:cut]])
    assert dec.frame_count(0) == kw["frames"] - 1  # the lone field waits
    dec.decode([b"\x00\x00\x00\x01\x0b"])           # end of stream
    assert dec.frame_count(0) == 1
    last = dec.read_frame_tight(0, 0, crop=False)
    dec.close()
    assert np.array_equal(last, ref[-1])
    y = last[:W * Hc].reshape(Hc, W)
    assert np.array_equal(y[0::2], rec[-1][:W * Hc].reshape(Hc, W)[0::2]) and (y[1::2] == 128).all()
    # the same in ONE chunk: the lone field goes out behind the complete frames
    dec = H.Decoder(max_streams=1, max_width=W, max_height=Hc, max_frames_per_batch=2 * kw["frames"], max_slices_per_frame=8)
    dec.decode([stream[:cut] + b"\x00\x00\x00\x01\x0a"])
    assert dec.frame_count(0) == kw["frames"]
    assert np.array_equal(dec.read_frames(0, crop=False), ref)
    dec.close()


def test_gpu_1080i_full_size(H, sg, oracle_mod):
    """1920x1080 interlaced (coded 1920x1088: 68 macroblock rows, 34 per field), every frame two field pictures, top field first, P fields with two
    reference frames = four reference fields, quarter-sample motion: GPU == generator == oracle, cropped output 1920x1080."""
    kw = dict(width=1920, height=1080, frames=3, idr_period=0, profile_idc=77, cabac=0, field_pics=1, num_ref_frames=2, motion_x4=5, motion_y4=-3, qp=30, seed=331)
    stream, rec, _ = sg.encode(**kw)
    out, info = _decode_gpu(H, [stream], 1920, 1080, 6, 1)
    assert info.n_frames == 3 and (info.width, info.height, info.coded_width, info.coded_height) == (1920, 1080, 1920, 1088)
    assert np.array_equal(out[0], rec), "GPU != generator reconstruction"
    ref, _ = oracle_mod.decode(stream, crop=False)
    assert np.array_equal(out[0], ref), "GPU != oracle"
    crop, _ = _decode_gpu(H, [stream], 1920, 1080, 6, 1, crop=True)
    y = rec[:, :1920 * 1088].reshape(-1, 1088, 1920)[:, :1080]
    assert np.array_equal(crop[0][:, :1920 * 1080].reshape(-1, 1080, 1920), y)
    # through the one-workgroup-per-picture kernels as well (what a launch of many streams uses)
    out0, _ = _decode_gpu(H, [stream], 1920, 1080, 6, 1, x_wgs=0)
    assert np.array_equal(out0[0], rec)


def test_gpu_field_and_frame_streams_share_a_batch(H, sg):
    """Field-coded, picture-adaptive and progressive streams side by side in one decoder: launches then mix pictures of 9 and of 4 macroblock rows,
    frame pitches and field pitches."""
    names = ["field_IP", "cabac_IPP", "field_mixed_paff", "field_b_spatial", "b_ibbp_cavlc", "field_high_8x8"]
    kws = [FULL_MATRIX[n] for n in names]
    enc = [sg.encode(**kw) for kw in kws]
    dec = H.Decoder(max_streams=len(kws), max_width=176, max_height=144, max_frames_per_batch=max(pictures_of(kw) for kw in kws), max_slices_per_frame=8)
    dec.decode([e[0] for e in enc])
    for i, (kw, e) in enumerate(zip(kws, enc)):
        W, Hc = (kw["width"] + 15) & ~15, (kw["height"] + 15) & ~15
        assert dec.frame_count(i) == kw["frames"], names[i]
        assert np.array_equal(dec.read_frames(i, crop=False, size=W * Hc * 3 // 2), e[1]), names[i]
    dec.close()


def test_gpu_cabac_field_pictures_on_request(H, sg, oracle_mod):
    """CABAC field pictures with h264mi_config.allow_unpinned_field_cabac = 1: all 24 field recipes coded with CABAC, GPU == oracle == generator in both
    kernel families (a picture inside one workgroup; banded), equal PicOrderCnt lists, the committed MD5s, and no slice counted as failed.  What this
    pins is the mechanism -- the field contexts' offsets, the field column of Table 9-43, the field scans under CABAC --: the context VALUES are unpinned
    (mi_cabac_mn.cpp), generator and oracle share them."""
    gold = json.load(open(os.path.join(os.path.dirname(GOLDEN), "field_cabac_md5.json")))
    assert set(gold) == set(FIELD_CABAC_MATRIX)
    for name in sorted(FIELD_CABAC_MATRIX):
        kw = FIELD_CABAC_MATRIX[name]
        stream, rec, _ = sg.encode(**kw)
        gen_pocs = sg.last_pocs().tolist()
        ref, _ = oracle_mod.decode(stream, crop=False)
        assert np.array_equal(ref, rec), name
        for x in (None, 0, 512):
            out, info = _decode_gpu(H, [stream], kw["width"], kw["height"], pictures_of(kw), _nslices(kw), x_wgs=x, allow_unpinned_field_cabac=1)
            assert np.array_equal(out[0], rec), (name, x)
            assert info.pocs[0] == gen_pocs, (name, x)
        assert hashlib.md5(out[0].tobytes()).hexdigest() == gold[name]["frames_md5"], name
    # the failure counter: a CABAC field slice cut short fails, is counted, and says why
    kw = FIELD_CABAC_MATRIX["field_IP_cabac"]
    stream, rec, _ = sg.encode(**dict(kw, slices=1))
    nals = H.read_nal_units(stream)
    last = nals[-1].Offset
    broken = stream[:last + 8] + bytes(len(stream) - last - 8)  # the last field: slice header kept, slice data zeroed
    dec = H.Decoder(max_streams=1, max_width=176, max_height=128, max_frames_per_batch=2 * kw["frames"], max_slices_per_frame=1, allow_unpinned_field_cabac=1)
    assert dec.unpinned_failures() == 0
    with pytest.raises(H.H264MIError) as e:
        dec.decode([broken])
    assert e.value.code == -8 and "unpinned" in str(e.value)
    assert dec.unpinned_failures() == 1
    dec.close()


def test_gpu_field_pictures_with_cabac_are_refused_with_a_reason(H, sg):
    """Field pictures with entropy_coding_mode_flag = 1 need the context initialisation values of field-coded blocks (ctxIdx 277-398, 436-459), which are
    UNPINNED in this tree: without h264mi_config.allow_unpinned_field_cabac they are refused (H264MI_EUNSUPPORTED) with a message that says so.  The stream:
    a CAVLC field stream whose PPS is patched to announce CABAC (the refusal comes at the first slice header, before any slice data is looked at)."""
    stream, _, _ = sg.encode(**FIELD_MATRIX["field_IP"])
    i = stream.find(b"\x00\x00\x01\x68") + 4  # pic_parameter_set_rbsp: ue(0) ue(0) entropy_coding_mode_flag ...
    assert i > 4 and stream[i] & 0xC0 == 0xC0 and not stream[i] & 0x20
    patched = stream[:i] + bytes([stream[i] | 0x20]) + stream[i + 1:]
    dec = H.Decoder(max_streams=1, max_width=176, max_height=128, max_frames_per_batch=10, max_slices_per_frame=8)
    with pytest.raises(H.H264MIError) as e:
        dec.decode([patched])
    assert e.value.code == -3 and "CABAC" in str(e.value) and "277" in str(e.value)
    dec.close()

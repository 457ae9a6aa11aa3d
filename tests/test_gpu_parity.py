"""GPU parity tests proper (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
must reproduce the oracle (and the generator's reconstruction) bit for bit."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import MATRIX

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "stream_md5.json")


def _decode_gpu(H, streams, w, h, frames, slices=1, crop=False):
    W, Hc = (w + 15) // 16 * 16, (h + 15) // 16 * 16
    dec = H.Decoder(max_streams=len(streams), max_width=W, max_height=Hc, max_frames_per_batch=frames, max_slices_per_frame=max(1, slices),
                    max_bitstream_bytes=sum(len(s) for s in streams) * 2 + (1 << 20))
    info = dec.decode(streams)
    size = (w * h if crop else W * Hc) * 3 // 2
    out = [dec.read_frames(i, crop=crop, size=size) for i in range(len(streams))]
    dec.close()
    return out, info


@pytest.mark.parametrize("name", sorted(MATRIX))
def test_gpu_matches_oracle_and_generator(name, H, sg, oracle_mod):
    kw = MATRIX[name]
    stream, rec, _ = sg.encode(**kw)
    ref, _ = oracle_mod.decode(stream, crop=False)
    out, info = _decode_gpu(H, [stream], kw["width"], kw["height"], kw["frames"], kw.get("slices", 1))
    assert info.n_frames == kw["frames"]
    assert out[0].shape == ref.shape
    assert np.array_equal(out[0], ref), "GPU != oracle"
    assert np.array_equal(out[0], rec), "GPU != generator reconstruction"


def test_gpu_golden_md5(H, sg):
    gold = json.load(open(GOLDEN))
    for name, g in sorted(gold.items()):
        kw = MATRIX[name]
        stream, _, _ = sg.encode(**kw)
        out, _ = _decode_gpu(H, [stream], kw["width"], kw["height"], kw["frames"], kw.get("slices", 1))
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


def test_gpu_corrupt_streams_do_not_hang_or_crash(H, sg):
    kw = dict(width=176, height=144, frames=3, idr_period=0, profile_idc=77, cabac=1, seed=5)
    stream, _, _ = sg.encode(**kw)
    rng = np.random.default_rng(0)
    for trial in range(6):
        b = bytearray(stream)
        for _ in range(8):
            i = int(rng.integers(60, len(b)))
            b[i] ^= 1 << int(rng.integers(0, 8))
        dec = H.Decoder(max_streams=1, max_width=176, max_height=144, max_frames_per_batch=3)
        try:
            dec.decode([bytes(b)])
        except H.H264MIError as e:
            assert e.code in (-2, -3, -8)
        dec.close()
    # truncated
    dec = H.Decoder(max_streams=1, max_width=176, max_height=144, max_frames_per_batch=3)
    try:
        dec.decode([stream[:len(stream) // 2]])
    except H.H264MIError as e:
        assert e.code in (-2, -8)
    dec.close()


def test_gpu_real_stream_matches_oracle(H, real_stream, oracle_mod):
    ref, info = oracle_mod.decode(real_stream, crop=True)
    out, _ = _decode_gpu(H, [real_stream], 320, 240, 36, crop=True)
    assert np.array_equal(out[0], ref)


def test_gpu_1080p_full_size_properties(H, sg, oracle_mod):
    """BASELINE config at full size: 1080p Main CABAC IPPP, cropping 1088 -> 1080, two streams."""
    kw = sg.recipe("C3", frames=4, idr_period=4)
    streams = [sg.encode(**dict(kw, seed=3 + i))[0] for i in range(2)]
    out, info = _decode_gpu(H, streams, 1920, 1080, 4, crop=True)
    assert (info.width, info.height, info.coded_width, info.coded_height) == (1920, 1080, 1920, 1088)
    for i in range(2):
        ref, _ = oracle_mod.decode(streams[i], crop=True)
        assert np.array_equal(out[i], ref)
    # device-side crop/pack kernel (K6) == host-side cropped read
    import torch
    dec = H.Decoder(max_streams=1, max_width=1920, max_height=1088, max_frames_per_batch=4)
    dec.decode([streams[0]])
    buf = torch.empty(1920 * 1080 * 3 // 2, dtype=torch.uint8, device="cuda")
    from h264decode_amd._lib import check
    check(dec._L.h264mi_frame_pack_device(dec._h, 0, 3, buf.data_ptr(), buf.numel()))
    dec.sync()
    assert np.array_equal(buf.cpu().numpy(), out[0][3])
    dec.close()


def test_gpu_720p_cavlc_intra(H, sg, oracle_mod):
    """BASELINE configs[1]: 720p Baseline CAVLC I-frames."""
    kw = sg.recipe("C2", frames=2)
    stream, rec, _ = sg.encode(**kw)
    out, _ = _decode_gpu(H, [stream], 1280, 720, 2)
    assert np.array_equal(out[0], rec)


def test_gpu_4k_high_8_slices(H, sg):
    """BASELINE configs[3]: 3840x2160 High CABAC, 8x8 transform, 8 slices per picture (240 x 135 macroblocks: widest row
    state, 34 deblocking row groups in 3 rounds, slice boundaries inside and across macroblock rows)."""
    kw = sg.recipe("C4", frames=3, idr_period=3)
    stream, rec, _ = sg.encode(**kw)
    out, info = _decode_gpu(H, [stream], 3840, 2160, 3, slices=8)
    assert (info.coded_width, info.coded_height) == (3840, 2160)
    assert np.array_equal(out[0], rec)


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


def test_gpu_c_program_through_the_abi(H, sg, oracle_mod, tmp_path):
    """examples/h264mi_decode.c (plain C, only include/h264mi.h): file in, raw I420 out, several batches."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")])
    stream = sg.encode(width=180, height=100, frames=8, idr_period=3, profile_idc=100, cabac=1, transform8x8=1, slices=2, long_start_code=0, seed=77)[0]
    ref, info = oracle_mod.decode(stream, crop=True)
    src, dst = tmp_path / "in.h264", tmp_path / "out.yuv"
    src.write_bytes(stream)
    subprocess.check_call([os.path.join(root, "examples", "h264mi_decode"), str(src), str(dst), "3"])
    got = np.frombuffer(dst.read_bytes(), dtype=np.uint8).reshape(-1, info.width * info.height * 3 // 2)
    assert np.array_equal(got, ref)


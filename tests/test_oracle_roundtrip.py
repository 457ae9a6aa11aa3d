"""CPU tests of the oracle: closed-loop round trips against the stream generator's independent
reconstruction, syntax coverage of the synthetic matrix, and a third-party real stream."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import MATRIX

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "stream_md5.json")


@pytest.mark.parametrize("name", sorted(MATRIX))
def test_roundtrip_oracle_equals_generator(name, sg, oracle_mod):
    """Bit-exact: oracle decode == generator reconstruction (two independently written recon paths)."""
    stream, rec, _ = sg.encode(**MATRIX[name])
    out, info = oracle_mod.decode(stream, crop=False)
    assert out.shape == rec.shape
    assert np.array_equal(out, rec)
    assert info.n_frames == MATRIX[name]["frames"]


def test_golden_md5(sg, oracle_mod):
    """Committed fixtures: MD5 of every synthetic stream and of its decoded frames
    (tests/golden/make_golden.py regenerates them)."""
    gold = json.load(open(GOLDEN))
    for name, g in gold.items():
        stream, _, _ = sg.encode(**MATRIX[name])
        assert hashlib.md5(stream).hexdigest() == g["stream_md5"], name
        out, _ = oracle_mod.decode(stream, crop=False)
        assert hashlib.md5(out.tobytes()).hexdigest() == g["frames_md5"], name


def test_matrix_exercises_the_syntax(sg, oracle_mod):
    """The matrix must actually contain every macroblock type / feature it claims to test."""
    seen = set()
    feats = set()
    for name, kw in MATRIX.items():
        stream, _, _ = sg.encode(**kw)
        _, info, tr = oracle_mod.decode(stream, crop=False, trace=True)
        cab = kw.get("cabac", 0)
        for raw, cbp, qp, mode, t8, mvx, mvy, ref in tr:
            if ref <= -100:
                continue  # macroblock of a B slice: checked below
            seen.add(("skip" if raw == -1 else int(raw), cab))
            if t8:
                feats.add(("t8x8", cab))
            if ref > 0:
                feats.add(("ref>0", cab))
            if mvx & 3 or mvy & 3:
                feats.add(("qpel", cab))
    # picture management: the generator must have emitted, and the oracle executed, every operation the matrix claims
    dpb = {0: 0, 1: 0}
    for name, kw in MATRIX.items():
        if not any(kw.get(k) for k in ("rplm", "mmco", "idr_long_term", "nonref_period", "slice_qp_delta", "fn_gap_period")) and kw.get("poc_type", 0) != 1:
            continue
        stream, _, _ = sg.encode(**kw)
        emitted = sg.last_features()
        oracle_mod.decode(stream, crop=False)
        assert oracle_mod.last_features == emitted, (name, hex(emitted), hex(oracle_mod.last_features))
        assert np.array_equal(oracle_mod.last_pocs, sg.last_pocs()), name
        dpb[kw.get("cabac", 0)] |= emitted
    for cab in (0, 1):
        for bit, what in [(1, "mmco1"), (2, "mmco2"), (3, "mmco3"), (4, "mmco4"), (6, "mmco6"), (8, "rplm idc0"), (9, "rplm idc1"),
                          (10, "rplm idc2"), (11, "long-term ref in list"), (12, "non-ref picture"), (13, "slice_qp_delta"),
                          (15, "frame_num gap filled with non-existing frames")]:
            assert dpb[cab] >> bit & 1, (what, cab)
    assert dpb[1] >> 5 & 1 and dpb[1] >> 14 & 1  # mmco5 and POC type 1 with a non-zero delta (CABAC cases)
    for cab in (0, 1):
        for raw in ("skip", 0, 1, 2, 3, 25, 30):  # P16x16/I_NxN, 16x8/I16, 8x16, P8x8, I_PCM (I and P slices)
            assert (raw, cab) in seen, (raw, cab)
        for f in ("t8x8", "ref>0", "qpel"):
            assert (f, cab) in feats, (f, cab)
    # B pictures: every inter mb_type of Table 7-14 (0..22), intra types inside B slices (23.., incl. I_PCM = 48), B_Skip
    bseen = {0: set(), 1: set()}
    for name, kw in MATRIX.items():
        if not kw.get("bframes"):
            continue
        stream, _, _ = sg.encode(**kw)
        _, info, tr = oracle_mod.decode(stream, crop=False, trace=True)
        bseen[kw.get("cabac", 0)] |= {int(r[0]) for r in tr[tr[:, 7] <= -100]}
    for cab in (0, 1):
        assert set(range(23)) <= bseen[cab] and -1 in bseen[cab] and any(t >= 23 for t in bseen[cab]), (cab, sorted(bseen[cab]))
    assert 48 in bseen[1]


def test_crop(sg, oracle_mod):
    stream, rec, _ = sg.encode(width=180, height=100, frames=2, idr_period=0, profile_idc=77, cabac=1)
    out, info = oracle_mod.decode(stream, crop=True)
    assert (info.width, info.height, info.coded_width, info.coded_height) == (180, 100, 192, 112)
    full = rec[0][:192 * 112].reshape(112, 192)
    assert np.array_equal(out[0][:180 * 100].reshape(100, 180), full[:100, :180])


def test_truncated_stream_is_an_error_not_a_crash(sg, oracle_mod):
    stream, _, _ = sg.encode(width=64, height=48, frames=2, idr_period=0, profile_idc=77, cabac=1)
    for cut in (len(stream) // 2, len(stream) - 7, 40):
        try:
            oracle_mod.decode(stream[:cut], crop=False)
        except oracle_mod.OracleError:
            pass  # an error is fine; a crash is not


def test_empty_stream(oracle_mod):
    out, info = oracle_mod.decode(b"", crop=False)
    assert info.n_frames == 0


def test_real_x264_stream_self_synchronises(real_stream, oracle_mod):
    """Third-party High-profile CABAC stream (x264, 8x8 transform, deblocking): every slice must decode
    exactly PicSizeInMbs macroblocks and end on end_of_slice_flag -- any error in the context tables,
    binarisations or ctxIdxInc rules desynchronises the arithmetic decoder long before that."""
    out, info = oracle_mod.decode(real_stream, crop=True)
    assert (info.width, info.height) == (320, 240)
    assert info.n_frames == 36
    assert info.n_mbs == 36 * 300
    # drift check: with a wrong transform / MC / deblock the P-frame chain diverges visibly
    y = out[:, :320 * 240].astype(np.int32)
    step = np.abs(y[1:] - y[:-1]).mean(axis=1)
    assert step.max() < 12.0, step
    # blockiness: mean |gradient| across 16-pixel MB boundaries must not exceed the interior gradient by much
    last = y[-1].reshape(240, 320)
    gx = np.abs(np.diff(last, axis=1))
    on_edge = gx[:, 15::16].mean()
    interior = np.delete(gx, np.s_[15::16], axis=1).mean()
    assert on_edge < 1.5 * interior + 1.0, (on_edge, interior)


def test_extension_nal_units_are_passed_over(sg, oracle_mod):
    """An SVC / MVC stream is its base layer / base view plus NAL units of types 14, 15, 20, 21: the checker decodes the base and ignores the rest."""
    from conftest import with_extension_nals
    for kw in (dict(width=176, height=144, frames=6, idr_period=3, profile_idc=77, cabac=1, slices=2, seed=41),
               dict(width=176, height=144, frames=7, idr_period=0, profile_idc=100, cabac=0, transform8x8=1, bframes=2, num_ref_frames=3, seed=42)):
        stream, rec, _ = sg.encode(**kw)
        ext = with_extension_nals(stream, seed=kw["seed"])
        assert len(ext) > len(stream)
        out, _ = oracle_mod.decode(ext, crop=False)
        assert np.array_equal(out, rec)

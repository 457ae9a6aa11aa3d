"""The product's picture management on the CPU: mi_api.cpp + mi_parse.cpp built against the null device of tools/hoststub (kernels are
not run) prepare every matrix stream, and the PicOrderCnt / frame_num / IDR flag they report per picture (h264mi_frame_get_info) must
be the generator's -- 8.2.1 for all three pic_order_cnt_types, memory management operation 5, frame_num gaps, B pictures in coding
order, non-reference pictures, and a bottom field that is not at the top field's count (POC_MATRIX).  The oracle is held to the same
numbers.  This is the product's own host code, not a restatement: what it cannot show is anything the kernels do."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import FULL_MATRIX, MATRIX, POC_MATRIX, pictures_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not shutil.which("g++"), reason="g++ not available")


@pytest.fixture(scope="module")
def host_pocs(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("host_pocs")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "host_pocs.sh")], env=dict(os.environ, TMPDIR=str(tmp)), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout.strip().splitlines()[-1], tmp


def _host(prog, tmp, stream, kw):
    path = os.path.join(str(tmp), "s.h264")
    open(path, "wb").write(stream)
    W, Hc = (kw["width"] + 15) & ~15, (kw["height"] + 15) & ~15
    nsl = max(1, kw.get("slices", 1)) * max(1, kw.get("slice_groups", 1))
    r = subprocess.run([prog, path, str(W), str(Hc), str(pictures_of(kw)), str(max(8, nsl))], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [line.split() for line in r.stdout.splitlines() if line.strip()]
    return np.array([[int(x) for x in t] for t in lines if t[0] != "order"]), [int(x) for t in lines if t[0] == "order" for x in t[1:]]


def test_host_picture_order_counts_match_the_generator(host_pocs, sg):
    prog, tmp = host_pocs
    for name, kw in sorted(FULL_MATRIX.items()):  # (field pictures included: second-field detection, field counts, the lists and marking of 8.2.4.2.5 / 8.2.5.4.1)
        stream, _, _ = sg.encode(want_recon=False, **kw)
        want = sg.last_pocs()
        got, order = _host(prog, tmp, stream, kw)
        assert got.shape == (kw["frames"], 5), name
        assert np.array_equal(got[:, 0], want), (name, got[:, 0], want)
        assert got[0, 3] == 1 and got[0, 4] == 1, name  # the first picture is an IDR picture and starts a sequence
        # h264mi_stream_output_order: a permutation that shows every coded video sequence in ascending PicOrderCnt, the sequences in coding order
        assert sorted(order) == list(range(kw["frames"])), name
        seq = np.cumsum(got[:, 4])
        keys = [(int(seq[i]), int(got[i, 0])) for i in order]
        assert keys == sorted(keys), (name, keys)
        # a new sequence exactly where the generator's counts start over: IDR pictures and pictures with operation 5
        if not kw.get("bframes") and kw.get("poc_type", 0) != 2 and not kw.get("poc_bottom_delta"):
            restarts = np.flatnonzero(np.diff(want) < 0) + 1
            assert set(restarts.tolist()) <= set(np.flatnonzero(got[:, 4]).tolist()), name


def test_host_passes_over_extension_nal_units(host_pocs, sg):
    """The base layer / base view of an SVC / MVC stream: the product's host code sees the same pictures with and without the NAL units of types 14, 15, 20, 21
    (and SEI, filler data) between them."""
    from conftest import with_extension_nals
    prog, tmp = host_pocs
    for name in ("cabac_IPP", "b_pyramid_cabac", "field_mixed_paff", "fmo_dispersed_aso"):
        kw = FULL_MATRIX[name]
        stream, _, _ = sg.encode(want_recon=False, **kw)
        want = sg.last_pocs()
        plain, order = _host(prog, tmp, stream, kw)
        got, order2 = _host(prog, tmp, with_extension_nals(stream, seed=7), kw)
        assert np.array_equal(got, plain) and order2 == order, name
        assert np.array_equal(got[:, 0], want), name


@pytest.mark.parametrize("name", sorted(POC_MATRIX))
def test_bottom_field_counts_oracle_equals_generator(name, sg, oracle_mod):
    kw = POC_MATRIX[name]
    stream, rec, _ = sg.encode(**kw)
    out, info = oracle_mod.decode(stream, crop=False)
    assert np.array_equal(out, rec)
    assert np.array_equal(oracle_mod.last_pocs, sg.last_pocs())
    # the syntax really is there: the PPS flag, and a non-zero delta in the slice headers of non-IDR pictures
    import h264decode_amd.h264 as H
    nals = H.read_nal_units(stream)
    sps = H.NewSPS(nals[0].RBSP())
    pps = H.NewPPS(sps, nals[1].RBSP())
    assert pps.BottomFieldPicOrderInFramePresent
    vs = H.VideoStream(sps, pps)
    deltas = set()
    for n in nals[2:]:
        if n.Type == 1:
            h = H.NewSliceContext(vs, n, n.RBSP()).Slice.Header
            deltas.add(h.DeltaPicOrderCntBottom if kw.get("poc_type", 0) == 0 else h.DeltaPicOrderCnt[1] + 1)
    assert deltas == {kw["poc_bottom_delta"]}


def test_poc_golden_vectors(sg, oracle_mod):
    """Committed fixtures of the POC_MATRIX streams: MD5 of stream and decoded frames, and the PicOrderCnt list (tests/golden/make_golden.py)."""
    import hashlib
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "poc_md5.json")))
    assert set(gold) == set(POC_MATRIX)
    for name, g in gold.items():
        stream, _, _ = sg.encode(**POC_MATRIX[name])
        assert hashlib.md5(stream).hexdigest() == g["stream_md5"], name
        out, _ = oracle_mod.decode(stream, crop=False)
        assert hashlib.md5(out.tobytes()).hexdigest() == g["frames_md5"], name
        assert [int(x) for x in oracle_mod.last_pocs] == g["pocs"], name

"""Field pictures (PAFF): oracle == generator on streams whose frames are coded as two field pictures.

CPU only, and about the ORACLE: the product refuses field pictures (H264MI_EUNSUPPORTED) until its kernels address the
frame store by field; what is pinned here is the checker the next step will be held against -- field views of the frame store,
the field reference lists of 8.2.4.2.5, the picture order counts of fields, the chroma vector offset between parities
(Table 8-9), the field scans, and the deblocking rules of field macroblocks."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import FIELD_CABAC_MATRIX, FIELD_MATRIX

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "field_md5.json")


ALL_FIELD = dict(FIELD_MATRIX, **FIELD_CABAC_MATRIX)


@pytest.mark.parametrize("name", sorted(ALL_FIELD))
def test_field_roundtrip_oracle_equals_generator(name, sg, oracle_mod):
    kw = ALL_FIELD[name]
    stream, rec, _ = sg.encode(**kw)
    out, info = oracle_mod.decode(stream, crop=False)
    assert info.n_frames == kw["frames"] and out.shape == rec.shape
    assert np.array_equal(out, rec)
    # one PicOrderCnt per output frame: the smaller of its two fields'
    assert np.array_equal(oracle_mod.last_pocs, sg.last_pocs())
    # cropped output: the display rectangle of the woven frame
    W, Hc = (kw["width"] + 15) & ~15, (kw["height"] + 15) & ~15
    if (W, Hc) != (kw["width"], kw["height"]):
        crop, cinfo = oracle_mod.decode(stream, crop=True)
        assert (cinfo.width, cinfo.height) == (kw["width"], kw["height"])
        y = rec[:, :W * Hc].reshape(-1, Hc, W)[:, :kw["height"], :kw["width"]]
        assert np.array_equal(crop[:, :kw["width"] * kw["height"]].reshape(-1, kw["height"], kw["width"]), y)


def test_field_golden_md5(sg, oracle_mod):
    gold = json.load(open(GOLDEN))
    assert set(gold) == set(FIELD_MATRIX)
    gold.update(json.load(open(os.path.join(os.path.dirname(GOLDEN), "field_cabac_md5.json"))))
    assert set(gold) == set(ALL_FIELD)
    for name, g in gold.items():
        stream, _, _ = sg.encode(**ALL_FIELD[name])
        assert hashlib.md5(stream).hexdigest() == g["stream_md5"], name
        out, _ = oracle_mod.decode(stream, crop=False)
        assert hashlib.md5(out.tobytes()).hexdigest() == g["frames_md5"], name


def test_field_slice_headers(H, sg):
    """h264/slice.go:867-872 (field_pic_flag, bottom_field_flag): the host parser reads the field flags; the two fields of a frame share
    frame_num, only the first field of an IDR frame is an IDR picture, and the parities alternate in the order the recipe asks for."""
    for name in ("field_IP", "field_bottom_first"):
        kw = FIELD_MATRIX[name]
        stream, _, _ = sg.encode(**kw)
        nals = H.read_nal_units(stream)
        sps = H.NewSPS(nals[0].RBSP())
        assert not sps.FrameMbsOnly and not sps.MbAdaptiveFrameField
        vs = H.VideoStream(sps, H.NewPPS(sps, nals[1].RBSP()))
        fields = []
        for n in nals[2:]:
            if n.Type in (1, 5):
                h = H.NewSliceContext(vs, n, n.RBSP()).Slice.Header
                assert h.FieldPic
                fields.append((n.Type, h.FrameNum, int(h.BottomField)))
        assert len(fields) == 2 * kw["frames"]
        first_parity = 1 if kw["field_pics"] == 2 else 0
        for t in range(kw["frames"]):
            a, b = fields[2 * t], fields[2 * t + 1]
            assert a[1] == b[1] and a[2] == first_parity and b[2] == 1 - first_parity
            assert b[0] == 1 and a[0] == (5 if t == 0 else 1)


def test_single_field_at_the_end(sg, oracle_mod):
    """A first field whose second field never comes still goes out as a frame: its rows decoded, the other parity left at the
    grey the frame store starts with."""
    kw = dict(FIELD_MATRIX["field_IP"], slices=1)
    stream, rec, _ = sg.encode(**kw)
    cut = stream.rfind(b"\x00\x00\x01")  # the last NAL unit = the second field of the last frame
    cut -= 1 if stream[cut - 1] == 0 else 0
    out, info = oracle_mod.decode(stream[:cut], crop=False)
    assert info.n_frames == kw["frames"]
    assert np.array_equal(out[:-1], rec[:-1])
    W, Hc = (kw["width"] + 15) & ~15, (kw["height"] + 15) & ~15
    y, ry = out[-1][:W * Hc].reshape(Hc, W), rec[-1][:W * Hc].reshape(Hc, W)
    assert np.array_equal(y[0::2], ry[0::2]) and (y[1::2] == 128).all()


def test_cabac_field_streams_use_the_field_contexts(sg, oracle_mod):
    """A CABAC field stream must differ from what the frame contexts would produce: decoding it as if its blocks were frame-coded (the oracle with the
    field flag of the residual parser forced off is not available, so the check is on the bit stream) -- the same recipe coded as frames and as
    fields shares no slice data; and Baseline (no CABAC, no interlace tools) stays refused by the generator."""
    kw = FIELD_CABAC_MATRIX["field_IP_cabac"]
    a, ra, _ = sg.encode(**kw)
    b, rb, _ = sg.encode(**dict(kw, field_pics=0))
    assert a != b and ra.shape == rb.shape
    out, _ = oracle_mod.decode(a, crop=False)
    assert np.array_equal(out, ra)
    with pytest.raises(RuntimeError, match="Main or High"):
        sg.encode(**dict(FIELD_MATRIX["field_IP"], profile_idc=66))


def test_access_units_of_field_streams(H, sg):
    """7.4.1.2.4: field_pic_flag / bottom_field_flag separate pictures -- the two fields of a frame are two access units even
    when every other header field agrees (pic_order_cnt_type 2: same frame_num, no POC syntax).  h264mi_slice_starts_picture
    (the decoder's own test) through the stream front-end's splitter."""
    for name in sorted(FIELD_MATRIX):
        kw = FIELD_MATRIX[name]
        stream, _, sizes = sg.encode(**kw)
        nals = H.read_nal_units(stream)
        sps = H.NewSPS(nals[0].RBSP())
        vs = H.VideoStream(sps, H.NewPPS(sps, nals[1].RBSP()))
        pictures, prev = 0, None
        for n in nals:
            if n.Type in (1, 5):
                h = H.NewSliceContext(vs, n, n.RBSP()).Slice.Header
                key = (h.FrameNum, bool(h.FieldPic), bool(h.BottomField), n.Type, n.RefIdc == 0)
                pictures += key != prev
                prev = key
        if kw["field_pics"] in (1, 2):
            assert pictures == 2 * kw["frames"]
        for piece in (1 << 20, 97):  # whatever the feeding pattern
            sp = H.AccessUnitSplitter(max_units_per_chunk=1)
            aus = []
            for i in range(0, len(stream), piece):
                aus += sp.feed(stream[i:i + piece])
            aus += sp.flush()
            assert len(aus) == pictures and b"".join(aus) == stream, (name, piece)
            # every frame of the generator (one or two pictures) ends where an access unit ends
            ends = set(np.cumsum([len(a) for a in aus]).tolist())
            assert set(np.cumsum(sizes).tolist()) <= ends, (name, piece)


def test_slice_group_maps_of_field_pictures(H, sg, oracle_mod):
    """8.2.2.8 (h264/slice.go:134-158): in a field picture a map unit is one macroblock, in a frame picture of the same interlace
    stream two macroblock rows.  The PRODUCT's h264mi_mb_to_slice_group_map (field_pic = 1) against the oracle's, for every slice
    of the slice-group field recipes -- host code that is ready before the kernels are."""
    import ctypes
    from oracle import lib as olib
    O = olib()
    O.h264o_parse_pps_ids.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    O.h264o_parse_sps.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    O.h264o_mb_to_slice_group_map.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    seen = set()
    for name, kw in FIELD_MATRIX.items():
        if not kw.get("slice_groups"):
            continue
        stream, _, _ = sg.encode(**kw)
        nals = H.read_nal_units(stream)
        sps = H.NewSPS(nals[0].RBSP())
        pps = H.NewPPS(sps, nals[1].RBSP())
        osps = ctypes.create_string_buffer(O.h264o_sizeof_sps() * 32)
        opps = ctypes.create_string_buffer(O.h264o_sizeof_pps())
        assert O.h264o_parse_sps(nals[0].RBSP(), len(nals[0].RBSP()), osps) == 0
        oids = np.zeros(1 << 16, dtype=np.uint8)
        n_ids = ctypes.c_size_t(0)
        assert O.h264o_parse_pps_ids(nals[1].RBSP(), len(nals[1].RBSP()), osps, opps, oids.ctypes.data, oids.size, ctypes.byref(n_ids)) == 0
        vs = H.VideoStream(sps, pps)
        frame_mbs = (sps.PicWidthInMbsMinus1 + 1) * (sps.PicHeightInMapUnitsMinus1 + 1) * 2
        for n in nals[2:]:
            if n.Type not in (1, 5):
                continue
            h = H.NewSliceContext(vs, n, n.RBSP()).Slice.Header
            m = H.MbToSliceGroupMap(sps, pps, h)
            assert m.size == (frame_mbs // 2 if h.FieldPic else frame_mbs)
            om = np.zeros(m.size, dtype=np.uint8)
            assert O.h264o_mb_to_slice_group_map(osps, opps, oids.ctypes.data, h.SliceGroupChangeCycle, int(bool(h.FieldPic)), om.ctypes.data) == m.size
            assert np.array_equal(m, om), name
            assert len(set(m.tolist())) > 1 or 3 <= kw["fmo_type"] <= 5
            seen.add((kw["fmo_type"], bool(h.FieldPic)))
    assert {(1, True), (3, True), (3, False), (6, True)} <= seen

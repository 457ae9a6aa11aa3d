"""CPU tests of the product's host layer (libh264mi.so without a GPU): the C ABI loads and exports
every symbol the header declares, the reference-API mirror parses what the oracle parses, and errors
are status codes, never crashes.  No decode calls here (they need a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(H):
    hdr = open(os.path.join(ROOT, "include", "h264mi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(h264mi_\w+)\s*\(", hdr))
    assert len(declared) >= 20
    L = H.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "missing export " + name
    from h264decode_amd import _lib
    assert declared == set(_lib.EXPORTS)
    # ... and nothing else: no test hook, no C++ internal, no kernel stub (csrc/exports.map; the hooks live in libh264mi_hooks.so)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "h264decode_amd", "libh264mi.so")], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-2] in ("T", "D", "B", "R")}
    assert exported == declared, sorted(exported ^ declared)


def test_no_gpu_means_error_not_fallback(H):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(H.H264MIError) as ei:
        H.Decoder(max_streams=1, max_width=64, max_height=48, max_frames_per_batch=2)
    assert ei.value.code == -4  # H264MI_ENODEVICE


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "h264decode_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "h264o_", "libh264oracle", "oracle/_build", "oracle/_ref", "import streamgen"):
                    assert needle not in src, (needle, os.path.join(dirpath, f))


def test_annexb_scan_and_nal_parse(H, sg):
    stream, _, _ = sg.encode(width=64, height=48, frames=3, idr_period=0, profile_idc=77, cabac=1, long_start_code=0)
    nals = H.read_nal_units(stream)
    assert [n.Type for n in nals] == [7, 8, 5, 1, 1]
    assert all(n.ForbiddenZeroBit == 0 and n.RefIdc == 3 for n in nals)
    # 3- and 4-byte start codes, trailing zeros, leading garbage zeros
    raw = b"\x00\x00\x00" + b"\x00\x00\x01\x67\x42\x00" + b"\x00\x00\x00\x01\x68\xce\x00\x00" + b"\x00\x00\x01\x65\x88\x80"
    n2 = H.read_nal_units(raw)
    assert [(n.Type, n.NumBytes) for n in n2] == [(7, 2), (8, 2), (5, 3)]
    # emulation prevention: 00 00 03 xx -> 00 00 xx
    nu = H.NewNalUnit(bytes([0x65, 0x00, 0x00, 0x03, 0x01, 0xAA, 0x00, 0x00, 0x03, 0x00, 0x00, 0x03, 0x02]))
    assert nu.RBSP() == bytes([0x00, 0x00, 0x01, 0xAA, 0x00, 0x00, 0x00, 0x00, 0x02])
    assert H.read_nal_units(b"") == [] and H.read_nal_units(b"\x00\x00") == []


def test_real_sps_pps_known_fields(H):
    """The SPS/PPS of the third-party sample analysed by hand in SURVEY.md Appendix C."""
    sps = H.NewSPS(bytes.fromhex("640028ac2b40a0fd00f1226a"))
    assert (sps.Profile, sps.Level, sps.ChromaFormat) == (100, 40, 1)
    assert sps.Log2MaxFrameNumMinus4 == 4 and sps.PicOrderCountType == 2 and sps.MaxNumRefFrames == 1
    assert (sps.PicWidthInMbsMinus1 + 1, sps.PicHeightInMapUnitsMinus1 + 1) == (20, 15) and sps.FrameMbsOnly == 1
    assert (sps.width, sps.height) == (320, 240)
    pps = H.NewPPS(sps, bytes.fromhex("ee025cb0"))
    assert pps.EntropyCodingMode == 1 and pps.PicInitQpMinus26 == 9 and pps.DeblockingFilterControlPresent == 1 and pps.Transform8x8Mode == 1


def test_headers_match_oracle(H, sg, oracle_mod):
    """Every SPS / PPS / slice header field the product parses equals the oracle's value."""
    from oracle import lib as olib
    O = olib()

    class OSH(ctypes.Structure):
        pass
    for name, kw in (("hi", dict(width=180, height=100, frames=3, idr_period=0, profile_idc=100, cabac=1, transform8x8=1, slices=2, scaling_matrix=1)),
                     ("wp", dict(width=64, height=48, frames=4, idr_period=0, profile_idc=77, cabac=1, weighted_pred=1, num_ref_frames=2, poc_type=0))):
        stream, _, _ = sg.encode(**kw)
        nals = H.read_nal_units(stream)
        sps = H.NewSPS(nals[0].RBSP())
        pps = H.NewPPS(sps, nals[1].RBSP())
        assert sps.width == kw["width"] and sps.height == kw["height"]
        assert sps.PicWidthInMbsMinus1 == (kw["width"] + 15) // 16 - 1
        assert pps.Transform8x8Mode == kw.get("transform8x8", 0) and pps.WeightedPred == kw.get("weighted_pred", 0)
        if kw.get("scaling_matrix"):
            assert list(sps.scaling_list_4x4[0]) == [6, 13, 13, 20, 20, 20, 28, 28, 28, 28, 32, 32, 32, 37, 37, 42]
            assert list(sps.scaling_list_4x4[5]) == [10, 14, 14, 20, 20, 20, 24, 24, 24, 24, 27, 27, 27, 30, 30, 34]
            assert list(pps.scaling_list_8x8[1][:6]) == [9, 13, 13, 15, 13, 15]
        vs = H.VideoStream(sps, pps)
        fn = 0
        for n in nals[2:]:
            sc = H.NewSliceContext(vs, n, n.RBSP())
            h = sc.Slice.Header
            assert h.SliceType in (5, 7) and h.PPSID == 0
            assert h.slice_data_bit_offset > 0 and h.SliceQPy == 26 + pps.PicInitQpMinus26 + h.SliceQpDelta
            if n.Type == 5:
                fn = 0
            if h.FirstMbInSlice == 0 and n.Type != 5:
                fn += 1
            assert h.FrameNum == fn
            if kw.get("weighted_pred") and h.SliceType == 5:
                assert h.LumaLog2WeightDenom == 5 and h.ChromaLog2WeightDenom == 4


def test_b_slice_headers(H, sg, oracle_mod):
    """slice_header() of B slices (7.3.3): direct_spatial_mv_pred_flag, both num_ref_idx overrides, both list-modification
    flags, the list-1 half of pred_weight_table() -- the header must end exactly where the oracle starts slice_data()
    (checked through the picture count and PicOrderCnt list of a full oracle decode of the same stream)."""
    for kw in (dict(width=64, height=48, frames=8, idr_period=0, profile_idc=77, cabac=1, bframes=2, num_ref_frames=3, weighted_bipred=1, direct_temporal=1),
               dict(width=64, height=48, frames=8, idr_period=0, profile_idc=77, cabac=0, bframes=3, b_pyramid=1, weighted_bipred=2)):
        stream, _, _ = sg.encode(**kw)
        _, info = oracle_mod.decode(stream, crop=False)
        assert info.n_frames == kw["frames"]
        nals = H.read_nal_units(stream)
        sps = H.NewSPS(nals[0].RBSP())
        pps = H.NewPPS(sps, nals[1].RBSP())
        assert pps.WeightedBipred == kw["weighted_bipred"]
        vs = H.VideoStream(sps, pps)
        n_b = n_ref_b = 0
        pocs = []
        for n in nals[2:]:
            h = H.NewSliceContext(vs, n, n.RBSP()).Slice.Header
            pocs.append(h.PicOrderCntLsb)
            if h.SliceType % 5 != 1:
                continue
            n_b += 1
            n_ref_b += n.RefIdc != 0
            assert h.DirectSpatialMvPred == (0 if kw.get("direct_temporal") else 1)
            assert 0 <= h.NumRefIdxL1ActiveMinus1 <= 1 and 0 <= h.NumRefIdxL0ActiveMinus1 < sps.MaxNumRefFrames
            assert h.RefPicListModificationFlagL1 == 0 and h.slice_data_bit_offset > 0
            if kw["weighted_bipred"] == 1:
                assert h.LumaLog2WeightDenom == 5 and h.ChromaLog2WeightDenom == 4
                assert h.LumaWeightL1[0] == 32 and h.LumaOffsetL1[0] == 0  # the generator leaves entry 0 at its default
                assert all(-128 <= int(w) <= 127 for w in h.LumaWeightL1[:2])
            else:
                assert h.LumaLog2WeightDenom == 0  # no pred_weight_table() in the header
        assert n_b >= 3
        assert (n_ref_b > 0) == bool(kw.get("b_pyramid"))
        assert [p // 2 for p in pocs][:4] == ([0, 3, 1, 2] if kw["bframes"] == 2 else [0, 4, 2, 1])  # coding order: anchor first, then its B pictures


def test_pred_weight_table_default_weight_128_is_accepted(H, sg):
    """7.4.3.2: only CODED weights are limited to -128..127.  With luma_log2_weight_denom = 7 an entry whose flag is 0 infers
    the weight 2^7 = 128, which the range check must not refuse (P slices, and list 1 of B slices)."""
    for kw, is_b in ((dict(width=64, height=48, frames=5, idr_period=0, profile_idc=77, cabac=0, weighted_pred=2, num_ref_frames=3), False),
                     (dict(width=64, height=48, frames=8, idr_period=0, profile_idc=77, cabac=1, bframes=2, num_ref_frames=3, weighted_bipred=1, weighted_pred=2), True)):
        stream, _, _ = sg.encode(**kw)
        nals = H.read_nal_units(stream)
        sps = H.NewSPS(nals[0].RBSP())
        pps = H.NewPPS(sps, nals[1].RBSP())
        assert pps.WeightedPred == 1
        vs = H.VideoStream(sps, pps)
        seen = 0
        for n in nals[2:]:
            h = H.NewSliceContext(vs, n, n.RBSP()).Slice.Header  # raises H264MIError if the header is refused
            st = h.SliceType % 5
            if st == 2:
                continue
            if st == 0 or (st == 1 and is_b):
                assert h.LumaLog2WeightDenom == 7 and h.ChromaLog2WeightDenom == 7
                if st == 0:  # the generator leaves entry 0 of list 0 (P slices) / of list 1 (B slices) at the inferred default
                    assert h.LumaWeightL0[0] == 128 and h.LumaOffsetL0[0] == 0
                    if h.NumRefIdxL0ActiveMinus1 >= 1:
                        assert 119 <= h.LumaWeightL0[1] <= 127
                else:
                    assert h.LumaWeightL1[0] == 128 and 119 <= h.LumaWeightL0[0] <= 127
                seen += 1
        assert seen >= 2
    # a CODED weight of 128 stays an error, 127 is fine: hand-built P slice header with both denominators 7
    sps, pps = H.NewSPS(sg_sps_for_kat(H, sg)), None
    pps = H.NewPPS(sps, sg_pps_for_kat(H, sg))
    assert sps.Log2MaxFrameNumMinus4 == 4 and sps.PicOrderCountType == 0 and pps.DeblockingFilterControlPresent == 1

    def se(v):
        b = bin((2 * v - 1 if v > 0 else -2 * v) + 1)[2:]
        return "0" * (len(b) - 1) + b
    vs = H.VideoStream(sps, pps)
    for weight in (127, 128):
        bits = "1" + "00110" + "1" + "00000001" + "00000010"  # first_mb 0, slice_type 5 (P), pps 0, frame_num u(8) = 1, pic_order_cnt_lsb u(8) = 2
        bits += "0" + "0"                                      # num_ref_idx_active_override_flag, ref_pic_list_modification_flag_l0
        bits += "0001000" + "0001000"                          # luma / chroma log2_weight_denom = ue(7)
        bits += "1" + se(weight) + se(0) + "0"                 # luma_weight_l0_flag 1, weight, offset 0; chroma_weight_l0_flag 0
        bits += "0" + se(0) + "1"                              # adaptive_ref_pic_marking_mode_flag 0, slice_qp_delta 0, disable_deblocking_filter_idc ue(0)
        bits += se(0) + se(0) + "1"                            # alpha / beta offsets, then slice data
        raw = int(bits.ljust((len(bits) + 7) // 8 * 8, "0"), 2).to_bytes((len(bits) + 7) // 8, "big")
        if weight == 127:
            h = H.NewSliceContext(vs, H.NewNalUnit(bytes([0x41]) + raw), raw).Slice.Header
            assert h.LumaWeightL0[0] == 127 and h.LumaLog2WeightDenom == 7 and h.FrameNum == 1
        else:
            with pytest.raises(H.H264MIError):
                H.NewSliceContext(vs, H.NewNalUnit(bytes([0x41]) + raw), raw)


def sg_sps_for_kat(H, sg):
    stream, _, _ = sg.encode(width=64, height=48, frames=2, idr_period=0, profile_idc=77, cabac=0, weighted_pred=2, num_ref_frames=1)
    return H.read_nal_units(stream)[0].RBSP()


def sg_pps_for_kat(H, sg):
    stream, _, _ = sg.encode(width=64, height=48, frames=2, idr_period=0, profile_idc=77, cabac=0, weighted_pred=2, num_ref_frames=1)
    return H.read_nal_units(stream)[1].RBSP()


def test_malformed_inputs_return_status_codes(H):
    with pytest.raises(H.H264MIError):
        H.NewSPS(b"")
    with pytest.raises(H.H264MIError):
        H.NewSPS(b"\x64")
    sps = H.NewSPS(bytes.fromhex("640028ac2b40a0fd00f1226a"))
    with pytest.raises(H.H264MIError):
        H.NewPPS(sps, b"")
    # B slice: valid H.264 outside the implemented scope -> H264MI_EUNSUPPORTED (-3)
    pps = H.NewPPS(sps, bytes.fromhex("ee025cb0"))
    vs = H.VideoStream(sps, pps)
    nal = H.NewNalUnit(bytes([0x41]) + bytes([0b10100110, 0x00, 0x00]))  # first_mb 0, slice_type ue=1 (B)
    with pytest.raises(H.H264MIError) as ei:
        H.NewSliceContext(vs, nal, nal.RBSP())
    assert ei.value.code in (-3, -2)


def test_deblock_launch_plan_cannot_deadlock(H):
    """K5 hands rows from one group of 8 macroblock rows to the next through bounded LDS rings with back-pressure.  The
    groups of one round run side by side; the ring written by the LAST wavefront is read by wavefront 0 one round later, so
    it must hold a whole macroblock row (otherwise every wavefront would end up waiting for wavefront 0's previous group).
    The plan must fit the 160 KB of LDS and use as many wavefronts as there are groups, up to the kernel's limit."""
    import ctypes
    f = H.load_hooks().h264mi_internal_deblock_plan
    f.restype = ctypes.c_int32
    I32 = ctypes.c_int32
    f.argtypes = [I32, I32, ctypes.POINTER(I32), ctypes.POINTER(I32), ctypes.POINTER(I32), ctypes.POINTER(I32), ctypes.POINTER(ctypes.c_int64)]
    nw, ring, ring_last, nb, lds = I32(), I32(), I32(), I32(), ctypes.c_int64()
    assert f(8, 320, ctypes.byref(nw), ctypes.byref(ring), ctypes.byref(ring_last), ctypes.byref(nb), ctypes.byref(lds)) == 0
    maxw = nw.value  # the kernel's wavefront limit (LDS windows, register budget): 40 groups of a narrow picture use all of them
    assert 8 <= maxw <= 12
    wave_bytes = 9 * 1568 + 2 * 8 * 80  # nine windows + the DbPrm stage (mi_kernels.h: MI_DEBLOCK8_WAVE_BYTES)
    for wmb in list(range(1, 40)) + [45, 80, 120, 128, 240, 256, 300, 512]:
        for hmb in list(range(1, 80)) + [135, 136, 160, 320]:
            nw, ring, ring_last, nb, lds = I32(), I32(), I32(), I32(), ctypes.c_int64()
            assert f(wmb, hmb, ctypes.byref(nw), ctypes.byref(ring), ctypes.byref(ring_last), ctypes.byref(nb), ctypes.byref(lds)) == 0
            groups = (hmb + 7) // 8
            rounds = (groups + nw.value - 1) // nw.value
            assert 1 <= nw.value <= min(maxw, groups) and lds.value <= 160 * 1024, (wmb, hmb, nw.value, lds.value)
            assert lds.value == 512 + nw.value * wave_bytes + ((nw.value - 1) * ring.value + ring_last.value * nb.value) * 96
            assert nb.value == (2 if rounds > 2 else 1)  # a single whole-row buffer deadlocks from three rounds on (test_deblock_schedule_model)
            assert 1 <= ring.value <= wmb and ring.value >= min(wmb, 16)
            assert ring_last.value == (wmb if rounds > 1 else ring.value), (wmb, hmb, nw.value, ring.value, ring_last.value)
            if groups <= 8:
                assert nw.value == groups, (wmb, hmb, nw.value)  # one round whenever eight wavefronts are enough (with whole-row rings fewer fit)
    assert f(120, 68, ctypes.byref(nw), ctypes.byref(ring), ctypes.byref(ring_last), ctypes.byref(nb), ctypes.byref(lds)) == 0
    assert (nw.value, nb.value) == (8, 1)  # 1080p: eight wavefronts (two on a SIMD: the kernel's landing registers), the ninth group of four rows in a second round


class _Bits:
    """MSB-first bit writer for hand-assembled parameter sets."""

    def __init__(self):
        self.b = []

    def u(self, v, n):
        self.b += [(v >> (n - 1 - i)) & 1 for i in range(n)]
        return self

    def ue(self, v):
        v += 1
        n = v.bit_length()
        return self.u(0, n - 1).u(v, n) if n > 1 else self.u(1, 1)

    def se(self, v):
        return self.ue(2 * v - 1 if v > 0 else -2 * v)

    def bytes(self):
        bits = self.b + [1]
        bits += [0] * (-len(bits) % 8)
        return bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))


def _sps(log2_fn=4, poc_type=0, log2_poc=4, refs=1, wmb1=10, hmu1=8, crop=None, frame_mbs_only=1):
    w = _Bits().u(66, 8).u(0xC0, 8).u(30, 8).ue(0).ue(log2_fn).ue(poc_type)
    if poc_type == 0:
        w.ue(log2_poc)
    w.ue(refs).u(0, 1).ue(wmb1).ue(hmu1).u(frame_mbs_only, 1)
    if not frame_mbs_only:
        w.u(0, 1)
    w.u(1, 1)
    if crop:
        w.u(1, 1)
        for c in crop:
            w.ue(c)
    else:
        w.u(0, 1)
    return w.u(0, 1).bytes()


def test_sps_range_checks(H):
    """Fields that later size buffers, shift counts and addresses are range-checked at parse time (network-facing input):
    every violation is H264MI_EBITSTREAM (-2), every boundary value is accepted."""
    ok = H.NewSPS(_sps())
    assert (ok.width, ok.height) == (176, 144)
    assert H.NewSPS(_sps(log2_fn=12, log2_poc=12, refs=16, wmb1=511, hmu1=319)).PicWidthInMbsMinus1 == 511
    assert H.NewSPS(_sps(crop=(0, 87, 0, 71))).width == 2  # 176 - 2*87
    bad = [dict(log2_fn=13), dict(log2_poc=13), dict(poc_type=3), dict(refs=17), dict(wmb1=512), dict(hmu1=320),
           dict(wmb1=(1 << 31) - 2), dict(hmu1=(1 << 32) - 2), dict(wmb1=(1 << 32) - 2),
           dict(hmu1=160, frame_mbs_only=0),           # 2 * 161 macroblock rows
           dict(crop=(44, 44, 0, 0)), dict(crop=(0, 0, 36, 36)), dict(crop=((1 << 31), 0, 0, 0)), dict(crop=(0, 0, 0, (1 << 32) - 2))]
    for kw in bad:
        with pytest.raises(H.H264MIError) as ei:
            H.NewSPS(_sps(**kw))
        assert ei.value.code == -2, kw


def _sg_structs(H, W, Hm, ng, map_type, **f):
    from h264decode_amd import _lib
    sps, pps = _lib.Sps(), _lib.Pps()
    sps.pic_width_in_mbs_minus1, sps.pic_height_in_map_units_minus1, sps.frame_mbs_only = W - 1, Hm - 1, 1
    pps.num_slice_groups_minus1, pps.slice_group_map_type = ng - 1, map_type
    for k, v in f.items():
        if isinstance(v, (list, tuple)):
            for i, x in enumerate(v):
                getattr(pps, k)[i] = x
        else:
            setattr(pps, k, v)
    return sps, pps


def _sg_map(H, sps, pps, cycle=0, ids=None):
    L = H.load()
    n = (sps.pic_width_in_mbs_minus1 + 1) * (sps.pic_height_in_map_units_minus1 + 1)
    out = np.zeros(n, dtype=np.uint8)
    ids = np.asarray(ids, dtype=np.uint8) if ids is not None else np.zeros(0, dtype=np.uint8)
    r = L.h264mi_map_unit_to_slice_group_map(ctypes.byref(sps), ctypes.byref(pps), ids.ctypes.data if ids.size else None, ids.size, cycle, out.ctypes.data, out.size, None)
    assert r == 0, r
    return out.reshape(sps.pic_height_in_map_units_minus1 + 1, -1)


def test_slice_group_maps_known_answers(H):
    """mapUnitToSliceGroupMap (8.2.2.1-8.2.2.7; h264/slice.go:457-529) on pictures small enough to work out by hand."""
    # type 0, interleaved: runs of 3 and 1 map units, wrapping over the row ends
    assert _sg_map(H, *_sg_structs(H, 4, 2, 2, 0, run_length_minus1=[2, 0])).ravel().tolist() == [0, 0, 0, 1, 0, 0, 0, 1]
    # type 1, dispersed: (x + ((y * n) >> 1)) % n
    assert _sg_map(H, *_sg_structs(H, 4, 3, 3, 1)).tolist() == [[0, 1, 2, 0], [1, 2, 0, 1], [0, 1, 2, 0]]
    # type 2, one foreground rectangle (1,0)-(2,1) and the left-over group
    assert _sg_map(H, *_sg_structs(H, 4, 3, 2, 2, top_left=[1], bottom_right=[6])).tolist() == [[1, 0, 0, 1], [1, 0, 0, 1], [1, 1, 1, 1]]
    # type 3, box-out clockwise from (2,2): left, up, right, right, down -- six units at change rate 1, cycle 6
    m = _sg_map(H, *_sg_structs(H, 4, 4, 2, 3, slice_group_change_direction=0, slice_group_change_rate_minus1=0), cycle=6)
    assert m.tolist() == [[1, 1, 1, 1], [1, 0, 0, 0], [1, 0, 0, 0], [1, 1, 1, 1]]
    # ... and counter-clockwise (direction flag 1) from (1,1): down, right, up, up(bounded) ...
    m = _sg_map(H, *_sg_structs(H, 4, 4, 2, 3, slice_group_change_direction=1, slice_group_change_rate_minus1=0), cycle=4)
    assert int((m == 0).sum()) == 4 and m[1][1] == 0 and m[2][1] == 0
    # type 4, raster: direction 1 puts the LAST cycle * rate units into group 0
    assert _sg_map(H, *_sg_structs(H, 4, 2, 2, 4, slice_group_change_direction=1, slice_group_change_rate_minus1=2), cycle=1).ravel().tolist() == [1, 1, 1, 1, 1, 0, 0, 0]
    # type 5, wipe: column by column
    assert _sg_map(H, *_sg_structs(H, 3, 2, 2, 5, slice_group_change_direction=0, slice_group_change_rate_minus1=2), cycle=1).tolist() == [[0, 0, 1], [0, 1, 1]]
    # type 6, explicit
    assert _sg_map(H, *_sg_structs(H, 3, 2, 3, 6, pic_size_in_map_units_minus1=5), ids=[2, 0, 1, 1, 0, 2]).ravel().tolist() == [2, 0, 1, 1, 0, 2]
    # the whole picture in group 0 once the cycle covers it; nothing when the cycle is 0
    assert not _sg_map(H, *_sg_structs(H, 4, 4, 2, 3, slice_group_change_rate_minus1=4), cycle=4).any()
    assert _sg_map(H, *_sg_structs(H, 4, 4, 2, 3, slice_group_change_rate_minus1=4), cycle=0).all()
    # nextMbAddress (8-17; h264/slice.go:530-552)
    L = H.load()
    m = np.array([0, 1, 1, 0, 2, 0], dtype=np.uint8)
    assert [L.h264mi_next_mb_address(m.ctypes.data, 6, n) for n in range(6)] == [3, 2, 6, 5, 6, 6]
    # type 6 without its slice_group_id array is an argument error, a small buffer a capacity error
    sps, pps = _sg_structs(H, 3, 2, 3, 6, pic_size_in_map_units_minus1=5)
    out = np.zeros(6, dtype=np.uint8)
    assert L.h264mi_map_unit_to_slice_group_map(ctypes.byref(sps), ctypes.byref(pps), None, 0, 0, out.ctypes.data, 6, None) == -1
    assert L.h264mi_map_unit_to_slice_group_map(ctypes.byref(sps), ctypes.byref(pps), None, 0, 0, out.ctypes.data, 5, None) != 0


def test_slice_group_syntax_and_maps_match_oracle(H, sg, oracle_mod):
    """PPS slice-group syntax (h264/pps.go:57-80), slice_group_change_cycle (h264/slice.go:1028-1031) and the macroblock-to-slice-group
    map (h264/slice.go:134-158) of generated streams: product == oracle, for every map type; the groups really are used."""
    from conftest import MATRIX
    from oracle import lib as olib
    O = olib()
    O.h264o_parse_pps_ids.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    O.h264o_parse_sps.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    O.h264o_mb_to_slice_group_map.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    types = set()
    for name, kw in MATRIX.items():
        if not kw.get("slice_groups"):
            continue
        stream, _, _ = sg.encode(**kw)
        nals = H.read_nal_units(stream)
        sps = H.NewSPS(nals[0].RBSP())
        pps = H.NewPPS(sps, nals[1].RBSP())
        assert pps.NumSliceGroupsMinus1 + 1 == (2 if 3 <= kw["fmo_type"] <= 5 else kw["slice_groups"]) and pps.SliceGroupMapType == kw["fmo_type"]
        types.add(pps.SliceGroupMapType)
        osps = ctypes.create_string_buffer(O.h264o_sizeof_sps() * 32)
        opps = ctypes.create_string_buffer(O.h264o_sizeof_pps())
        assert O.h264o_parse_sps(nals[0].RBSP(), len(nals[0].RBSP()), osps) == 0
        oids = np.zeros(1 << 16, dtype=np.uint8)
        n_ids = ctypes.c_size_t(0)
        assert O.h264o_parse_pps_ids(nals[1].RBSP(), len(nals[1].RBSP()), osps, opps, oids.ctypes.data, oids.size, ctypes.byref(n_ids)) == 0
        assert np.array_equal(oids[:n_ids.value], pps.SliceGroupId)
        vs = H.VideoStream(sps, pps)
        cycles = set()
        for n in nals[2:]:
            if n.Type not in (1, 5):
                continue
            h = H.NewSliceContext(vs, n, n.RBSP()).Slice.Header
            cycles.add(h.SliceGroupChangeCycle)
            m = H.MbToSliceGroupMap(sps, pps, h)
            om = np.zeros(m.size, dtype=np.uint8)
            assert O.h264o_mb_to_slice_group_map(osps, opps, oids.ctypes.data, h.SliceGroupChangeCycle, 0, om.ctypes.data) == m.size
            assert np.array_equal(m, om), name
            assert m[h.FirstMbInSlice] == m[H.nextMbAddress(h.FirstMbInSlice, sps, pps, h)] or H.nextMbAddress(h.FirstMbInSlice, sps, pps, h) == m.size
        if 3 <= kw["fmo_type"] <= 5:
            assert len(cycles) > 2, name  # the boundary moves from picture to picture
    assert types == set(range(7))

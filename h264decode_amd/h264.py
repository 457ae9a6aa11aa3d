"""Host-side mirror of the reference's Go package `h264` (same names, argument meaning and -- where
the reference is right -- values), implemented on the C ABI of libh264mi.so.

Reference            -> here
  NalUnit            h264/nalUnit.go:3-30      -> NalUnit (fields in CamelCase as in Go)
  NewNalUnit         h264/nalUnit.go:75        -> NewNalUnit(frame, numBytesInNal)
  (*NalUnit).RBSP    h264/nalUnit.go:72        -> NalUnit.RBSP()
  SPS / NewSPS       h264/sps.go:9,192         -> SPS / NewSPS(rbsp, showPacket)
  PPS / NewPPS       h264/pps.go:10,40         -> PPS / NewPPS(sps, rbsp, showPacket)
  SliceHeader        h264/slice.go:23          -> SliceHeader
  SliceContext       h264/slice.go:13          -> SliceContext (NalUnit, SPS, PPS, Slice.Header)
  NewSliceContext    h264/slice.go:835         -> NewSliceContext(videoStream, nalUnit, rbsp, showPacket)
  VideoStream        h264/slice.go:8           -> VideoStream(SPS, PPS, Slices)
  readNalUnit loop   h264/server.go:64-166     -> read_nal_units(bytes)
New (the reference has no pixel type): Decoder -- batched GPU decode to Y/Cb/Cr planes.

Error behaviour: the reference panics / os.Exit()s on malformed input (h264/server.go:136-143); here
every failure raises H264MIError carrying the C status code."""
import ctypes

import numpy as np

from . import _lib
from ._lib import H264MIError, check

NALU_TYPE_NAMES = {  # h264/frame.go:28-60
    0: "unspecified", 1: "coded slice of non-IDR picture", 2: "coded slice data partition A", 3: "coded slice data partition B",
    4: "coded slice data partition C", 5: "coded slice of an IDR picture", 6: "SEI", 7: "SPS", 8: "PPS", 9: "AUD",
    10: "end of sequence", 11: "end of stream", 12: "filler data", 13: "SPS extension", 14: "prefix NAL unit",
    15: "subset SPS", 19: "auxiliary slice", 20: "slice extension", 21: "slice extension for depth view"}


def _camel(name):
    return "".join(p.capitalize() if not p.isdigit() else p for p in name.split("_"))


class _Mirror:
    """Exposes the fields of a C struct under the reference's CamelCase names (and snake_case)."""
    _c = None

    def __getattr__(self, name):
        c = object.__getattribute__(self, "_c")
        for fname, _ in c._fields_:
            if name == fname or name == _camel(fname):
                v = getattr(c, fname)
                if hasattr(v, "_length_"):
                    return np.ctypeslib.as_array(v).copy()
                return v
        raise AttributeError(name)

    def as_dict(self):
        out = {}
        for fname, _ in self._c._fields_:
            v = getattr(self._c, fname)
            out[fname] = np.ctypeslib.as_array(v).tolist() if hasattr(v, "_length_") else v
        return out


class NalUnit(_Mirror):
    def __init__(self, c, rbsp):
        self._c, self._rbsp = c, rbsp

    def RBSP(self):
        return self._rbsp

    # Go field names that differ from the C struct's snake_case conversion
    NumBytes = property(lambda s: s._c.num_bytes)
    RefIdc = property(lambda s: s._c.ref_idc)
    Type = property(lambda s: s._c.type)


class SPS(_Mirror):
    def __init__(self, c):
        self._c = c

    ID = property(lambda s: s._c.id)  # h264/sps.go:12


class PPS(_Mirror):
    def __init__(self, c, slice_group_id=None):
        self._c = c
        self._ids = slice_group_id if slice_group_id is not None else np.zeros(0, dtype=np.uint8)

    SPSID = property(lambda s: s._c.sps_id)
    ID = property(lambda s: s._c.id)
    SliceGroupChangeDirection = property(lambda s: bool(s._c.slice_group_change_direction))
    SliceGroupId = property(lambda s: s._ids.copy())  # h264/pps.go:23 (slice_group_map_type 6: one entry per map unit)


class SliceHeader(_Mirror):
    def __init__(self, c):
        self._c = c

    PPSID = property(lambda s: s._c.pps_id)
    SliceQPy = property(lambda s: s._c.slice_qp_y)


class Slice:
    def __init__(self, header):
        self.Header, self.Data = header, None  # slice_data() is decoded on the GPU (Decoder)


class SliceContext:
    def __init__(self, nal, sps, pps, header):
        self.NalUnit, self.SPS, self.PPS, self.Slice = nal, sps, pps, Slice(header)


class VideoStream:
    def __init__(self, sps=None, pps=None):
        self.SPS, self.PPS, self.Slices = sps, pps, []


def NewNalUnit(frame: bytes, numBytesInNal: int = None) -> NalUnit:
    n = len(frame) if numBytesInNal is None else numBytesInNal
    L = _lib.load()
    c = _lib.Nal()
    rbsp = (ctypes.c_uint8 * max(n, 1))()
    rl = ctypes.c_size_t(0)
    check(L.h264mi_nal_parse(frame, n, ctypes.byref(c), rbsp, ctypes.byref(rl)))
    return NalUnit(c, bytes(rbsp[:rl.value]))


def NewSPS(rbsp: bytes, showPacket: bool = False) -> SPS:
    c = _lib.Sps()
    check(_lib.load().h264mi_sps_parse(rbsp, len(rbsp), ctypes.byref(c)))
    return SPS(c)


def NewPPS(sps: SPS, rbsp: bytes, showPacket: bool = False) -> PPS:
    c = _lib.Pps()
    L = _lib.load()
    check(L.h264mi_pps_parse(ctypes.byref(sps._c), rbsp, len(rbsp), ctypes.byref(c)))
    ids = None
    if c.num_slice_groups_minus1 > 0 and c.slice_group_map_type == 6:
        ids = np.zeros(c.pic_size_in_map_units_minus1 + 1, dtype=np.uint8)
        n = ctypes.c_size_t(0)
        check(L.h264mi_pps_slice_group_ids(ctypes.byref(sps._c), rbsp, len(rbsp), ids.ctypes.data, ids.size, ctypes.byref(n)))
    return PPS(c, ids)


def _sg_args(pps, header):
    ids = pps._ids
    cycle = header.SliceGroupChangeCycle if header is not None else 0
    return (ids.ctypes.data if ids.size else None), ids.size, int(cycle)


def MapUnitToSliceGroupMap(sps: SPS, pps: PPS, header=None) -> np.ndarray:
    """mapUnitToSliceGroupMap, 8.2.2.1-8.2.2.7 (h264/slice.go:457-529; the reference implements types 0-2)."""
    out = np.zeros((sps.PicWidthInMbsMinus1 + 1) * (sps.PicHeightInMapUnitsMinus1 + 1), dtype=np.uint8)
    ids, n_ids, cycle = _sg_args(pps, header)
    check(_lib.load().h264mi_map_unit_to_slice_group_map(ctypes.byref(sps._c), ctypes.byref(pps._c), ids, n_ids, cycle, out.ctypes.data, out.size, None))
    return out


def MbToSliceGroupMap(sps: SPS, pps: PPS, header=None) -> np.ndarray:
    """mbToSliceGroupMap, 8.2.2.8 (h264/slice.go:134-158)."""
    field = bool(header.FieldPic) if header is not None else False
    n = (sps.PicWidthInMbsMinus1 + 1) * (sps.PicHeightInMapUnitsMinus1 + 1) * (1 if (sps.FrameMbsOnly or field) else 2)
    out = np.zeros(n, dtype=np.uint8)
    ids, n_ids, cycle = _sg_args(pps, header)
    check(_lib.load().h264mi_mb_to_slice_group_map(ctypes.byref(sps._c), ctypes.byref(pps._c), ids, n_ids, cycle, int(field), out.ctypes.data, out.size, None))
    return out


def nextMbAddress(n: int, sps: SPS, pps: PPS, header=None) -> int:
    """(8-17), h264/slice.go:530-552: the next macroblock of n's slice group; PicSizeInMbs when there is none."""
    m = MbToSliceGroupMap(sps, pps, header)
    return int(_lib.load().h264mi_next_mb_address(m.ctypes.data, m.size, n))


def NewSliceContext(videoStream: VideoStream, nalUnit: NalUnit, rbsp: bytes, showPacket: bool = False) -> SliceContext:
    c = _lib.SliceHdr()
    check(_lib.load().h264mi_slice_header_parse(ctypes.byref(videoStream.SPS._c), ctypes.byref(videoStream.PPS._c), nalUnit.RefIdc,
                                                nalUnit.Type, rbsp, len(rbsp), ctypes.byref(c)))
    return SliceContext(nalUnit, videoStream.SPS, videoStream.PPS, SliceHeader(c))


def read_nal_units(stream: bytes):
    """Annex-B scan: returns [NalUnit] (replaces the readNalUnit loop of h264/server.go:64-111,144-166)."""
    L = _lib.load()
    cap = 1024
    while True:
        arr = (_lib.Nal * cap)()
        n = ctypes.c_int32(0)
        r = L.h264mi_annexb_scan(stream, len(stream), arr, cap, ctypes.byref(n))
        if r == -7:
            cap *= 4
            continue
        check(r)
        break
    out = []
    for i in range(n.value):
        off, size = arr[i].offset, arr[i].num_bytes
        nu = NewNalUnit(stream[off:off + size], size)
        nu._c.offset = off
        out.append(nu)
    return out


class Decoder:
    """Batched GPU decoder: N independent Annex-B streams side by side on one MI355X."""

    def __init__(self, max_streams=1, max_width=1920, max_height=1088, max_frames_per_batch=32, max_slices_per_frame=8, device=0,
                 max_bitstream_bytes=0, hip_stream=None, max_ref_frames=0, coef_blocks_per_mb=0, b_pictures=0, allow_unpinned_field_cabac=0):
        L = _lib.load()
        cfg = _lib.Config()
        cfg.struct_size = ctypes.sizeof(cfg)
        cfg.b_pictures = b_pictures  # 1: the B-only buffers exist from the start (default: from the first B slice on)
        cfg.allow_unpinned_field_cabac = allow_unpinned_field_cabac  # 1: CABAC field pictures are decoded with the unpinned context tables of field-coded blocks (default: refused)
        cfg.device, cfg.max_streams, cfg.max_width, cfg.max_height = device, max_streams, max_width, max_height
        cfg.max_frames_per_batch, cfg.max_slices_per_frame, cfg.max_bitstream_bytes = max_frames_per_batch, max_slices_per_frame, max_bitstream_bytes
        cfg.hip_stream = hip_stream
        cfg.max_ref_frames, cfg.coef_blocks_per_mb = max_ref_frames, coef_blocks_per_mb  # 0: defaults (16 reference slots, 8 blocks per macroblock)
        self._h = ctypes.c_void_p()
        check(L.h264mi_decoder_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        self._L = L
        self.max_streams = max_streams
        self._keep = None

    def close(self):
        if self._h:
            self._L.h264mi_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        check(self._L.h264mi_decoder_set_stream(self._h, hip_stream))

    def reset(self):
        check(self._L.h264mi_decoder_reset(self._h))

    def reset_stream(self, stream):
        """Forget parameter sets, reference pictures and POC history of one stream slot (a new connection takes it over)."""
        check(self._L.h264mi_stream_reset(self._h, stream))

    def set_isolation(self, on=True):
        """With isolation a broken stream is dropped from the batch and marked instead of failing prepare() / sync()."""
        check(self._L.h264mi_decoder_set_isolation(self._h, int(on)))

    def stream_status(self, stream):
        st = ctypes.c_int32(0)
        check(self._L.h264mi_stream_status(self._h, stream, ctypes.byref(st)))
        return st.value

    def frame_info(self, stream, frame):
        """h264mi_frame_info of a decoded frame: display / coded size, crop origin, PicOrderCnt, frame_num, nal_ref_idc, idr."""
        fi = _lib.FrameInfo()
        check(self._L.h264mi_frame_get_info(self._h, stream, frame, ctypes.byref(fi)))
        return fi

    def device_bytes(self):
        """Device memory the decoder holds right now (h264mi_decoder_memory)."""
        n = ctypes.c_int64()
        check(self._L.h264mi_decoder_memory(self._h, ctypes.byref(n)))
        return int(n.value)

    def unpinned_failures(self):
        """Slices of CABAC field pictures that failed in the entropy kernels so far (h264mi_decoder_unpinned_failures)."""
        n = ctypes.c_int64()
        check(self._L.h264mi_decoder_unpinned_failures(self._h, ctypes.byref(n)))
        return n.value

    def coef_pool(self):
        """(used, capacity) of the residual-coefficient pool in 32-byte blocks (h264mi_decoder_coef_pool); call after sync()."""
        u, c = ctypes.c_int64(), ctypes.c_int64()
        check(self._L.h264mi_decoder_coef_pool(self._h, ctypes.byref(u), ctypes.byref(c)))
        return int(u.value), int(c.value)

    def set_profiling(self, on=True):
        check(self._L.h264mi_decoder_set_profiling(self._h, int(on)))

    def kernel_times_ms(self):
        a = (ctypes.c_double * 5)()
        check(self._L.h264mi_last_kernel_times(self._h, a))
        return dict(zip(("entropy", "inter", "intra", "deblock", "total"), list(a)))

    def launch_times_ms(self, kernel):
        """Duration of every launch of `kernel` ("entropy", "inter", "intra", "deblock") in the last profiled pass."""
        k = ("entropy", "inter", "intra", "deblock").index(kernel)
        n = ctypes.c_int32(0)
        check(self._L.h264mi_last_launch_times(self._h, k, None, 0, ctypes.byref(n)))
        a = (ctypes.c_float * max(n.value, 1))()
        check(self._L.h264mi_last_launch_times(self._h, k, a, n.value, ctypes.byref(n)))
        return list(a)[:n.value]

    def _args(self, streams):
        n = len(streams)
        bufs = (ctypes.c_void_p * n)()
        lens = (ctypes.c_size_t * n)()
        keep = []
        for i, s in enumerate(streams):
            if s:
                b = ctypes.create_string_buffer(bytes(s), len(s))
                keep.append(b)
                bufs[i] = ctypes.cast(b, ctypes.c_void_p)
                lens[i] = len(s)
        self._keep = keep
        return n, bufs, lens

    def prepare(self, streams):
        n, bufs, lens = self._args(streams)
        info = _lib.BatchInfo()
        check(self._L.h264mi_batch_prepare(self._h, n, bufs, lens, ctypes.byref(info)))
        self.info = info
        return info

    def execute(self):
        check(self._L.h264mi_batch_execute(self._h))

    def sync(self):
        check(self._L.h264mi_batch_sync(self._h))

    def decode(self, streams):
        """prepare + execute + sync; returns the batch info."""
        info = self.prepare(streams)
        self.execute()
        self.sync()
        return info

    def frame_count(self, stream=0):
        n = ctypes.c_int32(0)
        check(self._L.h264mi_stream_frame_count(self._h, stream, ctypes.byref(n)))
        return n.value

    def frame_planes(self, stream, frame):
        y, cb, cr = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        py, pc, w, h = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        check(self._L.h264mi_frame_device_planes(self._h, stream, frame, ctypes.byref(y), ctypes.byref(cb), ctypes.byref(cr), ctypes.byref(py),
                                                 ctypes.byref(pc), ctypes.byref(w), ctypes.byref(h)))
        return dict(y=y.value, cb=cb.value, cr=cr.value, pitch_y=py.value, pitch_c=pc.value, coded_width=w.value, coded_height=h.value)

    def read_frame(self, stream, frame, crop=True):
        p = self.frame_planes(stream, frame)
        buf = np.zeros(p["coded_width"] * p["coded_height"] * 3 // 2, dtype=np.uint8)
        check(self._L.h264mi_frame_read(self._h, stream, frame, int(crop), buf.ctypes.data, buf.nbytes))
        return buf

    def read_frame_tight(self, stream, frame, crop=True):
        """The frame as exactly w*h*3/2 bytes of its own geometry (display size when cropped)."""
        fi = self.frame_info(stream, frame)
        w, h = (fi.width, fi.height) if crop else (fi.coded_width, fi.coded_height)
        return self.read_frame(stream, frame, crop)[:w * h * 3 // 2]

    def read_frames(self, stream=0, crop=False, size=None):
        """All frames of `stream` from the last batch as uint8[n, w*h*3/2] (tight I420)."""
        n = self.frame_count(stream)
        out = []
        for f in range(n):
            b = self.read_frame(stream, f, crop)
            out.append(b if size is None else b[:size])
        return np.stack(out) if out else np.zeros((0, 0), np.uint8)

    def pack_batch(self, dst_ptr, cap, stream=-1):
        """K6 in one launch: every frame of the last executed batch (of one stream, or of all streams) cropped and packed
        back to back as tight I420 into the DEVICE buffer at dst_ptr.  Returns the number of bytes."""
        n = ctypes.c_size_t(0)
        check(self._L.h264mi_batch_pack_device(self._h, stream, dst_ptr, cap, ctypes.byref(n)))
        return n.value

    def output_order(self, stream=0):
        """Indices (decoding order) of the stream's frames of the last batch in display order: ascending PicOrderCnt per
        coded video sequence.  Only differs from range(n) for streams with B pictures."""
        n = self.frame_count(stream)
        order = (ctypes.c_int32 * max(n, 1))()
        got = ctypes.c_int32(0)
        check(self._L.h264mi_stream_output_order(self._h, stream, order, n, ctypes.byref(got)))
        return [order[i] for i in range(min(n, got.value))]

    def read_mbrecs(self, stream, frame, n_mbs):
        buf = np.zeros(n_mbs * 128, dtype=np.uint8)
        check(self._L.h264mi_frame_read_mbrecs(self._h, stream, frame, buf.ctypes.data, buf.nbytes))
        return buf.reshape(n_mbs, 128)

    def read_mbmv1(self, stream, frame, n_mbs):
        """List-1 motion vectors of a picture (int16 [n_mbs, 16, 2]; zeros for pictures without B slices)."""
        buf = np.zeros(n_mbs * 64, dtype=np.uint8)
        check(self._L.h264mi_frame_read_mbmv1(self._h, stream, frame, buf.ctypes.data, buf.nbytes))
        return buf.view(np.int16).reshape(n_mbs, 16, 2)


# ---------------------------------------------------------------------------------------------------
# Stream front-end: the reference's ByteStreamReader / handleConnection / readNalUnit
# (h264/server.go:64-172) read an endless Annex-B byte stream from a connection one byte at a time.
# Here the bytes are cut at access-unit boundaries and handed to the batched GPU decoder.

class DisplayOrder:
    """Output process for streams with B pictures: frames go in in decoding order, come out in display order.  A simplified
    C.4.5.3 "bumping": frames wait in a buffer ordered by PicOrderCnt; when more than `depth` frames wait the one with the
    smallest PicOrderCnt is released, and an IDR picture (or a picture with memory_management_control_operation 5: its
    PicOrderCnt is 0 again) releases everything before it.  `depth` must be at least the stream's number of reorder
    frames (consecutive B pictures + 1 for a B pyramid level); the SPS's max_num_ref_frames is always enough.
    (The reference has no output process: h264/server.go:113-166 stops at the parsed slice.)"""

    def __init__(self, depth=4):
        self.depth = depth
        self._wait = []  # (PicOrderCnt, arrival number, frame)
        self._n = 0

    def push(self, frame, pic_order_cnt, new_sequence=False):
        """Returns the frames that can be shown now, in display order."""
        out = []
        if new_sequence:
            out = self.flush()
        self._wait.append((pic_order_cnt, self._n, frame))
        self._n += 1
        self._wait.sort(key=lambda t: t[:2])
        while len(self._wait) > self.depth:
            out.append(self._wait.pop(0)[2])
        return out

    def flush(self):
        out = [t[2] for t in self._wait]
        self._wait = []
        return out


class AccessUnitSplitter:
    """Incremental Annex-B splitter: feed() arbitrary byte chunks, get back byte strings that end on an
    access-unit boundary (7.4.1.2.3/4): a new access unit starts at an access unit delimiter, at an SPS / PPS / SEI that follows
    a slice, or at the first slice of a new picture.  That last test is the decoder's own (h264mi_slice_starts_picture on the
    parsed slice headers, plus "a slice starts where a slice of this picture already started"): with slice groups or arbitrary
    slice order the slice of macroblock 0 is not the first of its picture, so `first_mb_in_slice == 0` -- what this class used
    to look at, and still does for slices whose parameter sets it has not seen -- would cut such pictures in the middle.
    3- and 4-byte start codes are accepted (Annex B.1); the tail that may still grow is held back until flush()."""

    def __init__(self, max_units_per_chunk=30):
        self._buf = bytearray()
        self._max = max(1, int(max_units_per_chunk))
        self._sps, self._pps = {}, {}  # parameter sets in force at the START of the buffer (those inside it are applied while scanning)

    @staticmethod
    def _nal_starts(buf):
        """offsets of the first start-code byte of every NAL unit"""
        out, i, n = [], 0, len(buf)
        while True:
            j = buf.find(b"\x00\x00\x01", i)
            if j < 0:
                return out
            out.append(j - 1 if j > 0 and buf[j - 1] == 0 else j)
            i = j + 3

    @staticmethod
    def _parameter_set(nal, sps, pps):
        """apply an SPS / PPS NAL (bytes from its header byte on) to the tables; unparsable ones are ignored"""
        try:
            nu = NewNalUnit(bytes(nal))
            if nu.Type == 7:
                s = NewSPS(nu.RBSP())
                sps[s.ID] = s
            elif nu.Type == 8 and len(nu.RBSP()) > 1:
                br = nu.RBSP()
                # seq_parameter_set_id is the second ue(v): parse against every known SPS until one accepts it (ids are small)
                for cand in list(sps.values()):
                    try:
                        q = NewPPS(cand, br)
                        if q.SPSID == cand.ID:
                            pps[q.ID] = q
                            break
                    except H264MIError:
                        continue
        except H264MIError:
            pass

    @staticmethod
    def _slice_header(nal, sps, pps):
        """(first_mb_in_slice, header, sps) of a slice NAL, or None if its parameter sets are unknown / it does not parse"""
        for n in (min(len(nal), 768), len(nal)):  # the header nearly always ends within the first bytes
            try:
                nu = NewNalUnit(bytes(nal[:n]))
                rb = nu.RBSP()
                # pic_parameter_set_id: third ue(v) of the header -- let every known PPS try (a stream rarely has more than one)
                for q in list(pps.values()):
                    sp = sps.get(q.SPSID)
                    if sp is None:
                        continue
                    try:
                        h = NewSliceContext(VideoStream(sp, q), nu, rb).Slice.Header
                        if h.PPSID == q.ID:
                            return h.FirstMbInSlice, h, sp
                    except H264MIError:
                        continue
            except H264MIError:
                pass
        return None

    def _boundaries(self, final):
        """offsets at which a new access unit starts (excluding 0), in order"""
        buf = self._buf
        starts = self._nal_starts(buf)
        sps, pps = dict(self._sps), dict(self._pps)
        cuts, seen_vcl = [], False
        first_hdr, first_mbs = None, set()
        L = _lib.load()
        for k, off in enumerate(starts):
            j = buf.find(b"\x00\x00\x01", off) + 3
            if j >= len(buf):
                break  # header byte not here yet
            end = starts[k + 1] if k + 1 < len(starts) else len(buf)
            complete = k + 1 < len(starts) or final
            t = buf[j] & 31
            if t in (1, 5):
                if not complete:
                    break  # the slice header may still be arriving
                parsed = self._slice_header(buf[j:end], sps, pps)
                if parsed is None:  # parameter sets unknown: the old rule
                    new_pic = j + 1 < len(buf) and (buf[j + 1] & 0x80) != 0
                    first_mb, hdr = (0 if new_pic else -1), None
                else:
                    first_mb, hdr, sp = parsed
                    new_pic = first_mb in first_mbs or (first_hdr is not None and L.h264mi_slice_starts_picture(ctypes.byref(sp._c), ctypes.byref(first_hdr._c), ctypes.byref(hdr._c)) == 1)
                if seen_vcl and new_pic:
                    cuts.append(off)
                    first_hdr, first_mbs = None, set()
                if first_hdr is None:
                    first_hdr = hdr
                first_mbs.add(first_mb)
                seen_vcl = True
            elif t in (6, 7, 8, 9):
                if seen_vcl:
                    cuts.append(off)
                    seen_vcl, first_hdr, first_mbs = False, None, set()
                if t in (7, 8) and complete:
                    self._parameter_set(buf[j:end], sps, pps)
            elif t in (10, 11) and seen_vcl:  # end of sequence / stream belong to the access unit they follow
                nxt = starts[k + 1] if k + 1 < len(starts) else None
                if nxt is not None:
                    cuts.append(nxt)
                    seen_vcl, first_hdr, first_mbs = False, None, set()
        # de-duplicate while keeping order
        out = []
        for c in cuts:
            if c > 0 and (not out or c > out[-1]):
                out.append(c)
        return out

    def feed(self, data: bytes):
        """Append bytes; returns a list of chunks, each holding whole access units (at most max_units_per_chunk)."""
        self._buf += data
        return self._emit(False)

    def flush(self):
        """End of stream: everything that is left is the last access unit(s)."""
        return self._emit(True)

    def _emit(self, final):
        cuts = self._boundaries(final)
        if final:
            cuts = cuts + [len(self._buf)]
        chunks, prev, k = [], 0, 0
        while k < len(cuts):
            take = min(self._max, len(cuts) - k)
            end = cuts[k + take - 1]
            if end > prev:
                chunks.append(bytes(self._buf[prev:end]))
            prev, k = end, k + take
        if prev:  # the parameter sets inside what leaves are in force for what stays
            head = self._buf[:prev]
            st = self._nal_starts(head)
            for k, off in enumerate(st):
                j = head.find(b"\x00\x00\x01", off) + 3
                if j < len(head) and (head[j] & 31) in (7, 8):
                    self._parameter_set(head[j:st[k + 1] if k + 1 < len(st) else len(head)], self._sps, self._pps)
        del self._buf[:prev]
        return chunks


END_OF_STREAM = b"\x00\x00\x00\x01\x0b"  # nal_unit_type 11: tells the decoder that nothing follows -- a first field still waiting for its second one goes out as it is


class H264Reader:
    """Mirror of the reference's H264Reader + handleConnection loop (h264/server.go:113-166): reads a connection
    (anything with recv() or read()), cuts the byte stream at access units and decodes them on the GPU.
    `on_frames(frames)` receives uint8[n, width*height*3/2] arrays (cropped I420) in decoding order, or -- with
    display_order=N -- in display order through a DisplayOrder buffer of depth N (streams with B pictures)."""

    def __init__(self, connection, decoder=None, on_frames=None, max_width=1920, max_height=1088, frames_per_batch=30, read_size=1 << 16, display_order=0):
        self.Stream = connection
        self.frames_per_batch = frames_per_batch
        self.decoder = decoder or Decoder(max_streams=1, max_width=(max_width + 15) // 16 * 16, max_height=(max_height + 15) // 16 * 16,
                                          max_frames_per_batch=frames_per_batch, max_slices_per_frame=16)
        self.on_frames = on_frames
        self.read_size = read_size
        self.splitter = AccessUnitSplitter(frames_per_batch)
        self.n_frames = 0
        self._dims = (0, 0)
        self._reorder = DisplayOrder(display_order) if display_order else None

    def _read(self):
        s = self.Stream
        return s.recv(self.read_size) if hasattr(s, "recv") else s.read(self.read_size)

    def _decode(self, chunks):
        for c in chunks:
            self.decoder.decode([c])
            n = self.decoder.frame_count(0)
            if n:
                frames = [self.decoder.read_frame(0, f, crop=True)[:self._size()] for f in range(n)]
                self.n_frames += n
                if self._reorder:
                    shown = []
                    for f in range(n):
                        fi = self.decoder.frame_info(0, f)
                        shown += self._reorder.push(frames[f], fi.pic_order_cnt, bool(fi.new_sequence))
                    frames = shown
                if self.on_frames and len(frames):
                    self.on_frames(np.stack(frames))

    def _size(self):
        info = self.decoder.info
        if info.width and info.height:
            self._dims = (info.width, info.height)
        return self._dims[0] * self._dims[1] * 3 // 2

    def run(self):
        """Read until the peer closes the connection; returns the number of decoded frames."""
        while True:
            data = self._read()
            if not data:
                break
            self._decode(self.splitter.feed(data))
        self._decode(self.splitter.flush() + [END_OF_STREAM])
        if self._reorder:
            rest = self._reorder.flush()
            if self.on_frames and rest:
                self.on_frames(np.stack(rest))
        return self.n_frames


def handleConnection(connection, **kw):
    """h264/server.go:113 -- decode everything arriving on `connection`; returns the frame count."""
    return H264Reader(connection, **kw).run()


def ByteStreamReader(connection, **kw):
    """h264/server.go:168 -- same, and closes the connection afterwards."""
    try:
        return handleConnection(connection, **kw)
    finally:
        if hasattr(connection, "close"):
            connection.close()


class BatchServer:
    """Several connections decoded side by side: every connection owns one stream slot of ONE batched Decoder, and each
    tick hands one chunk (whole access units) per connection that has one to a single h264mi_decode_batch call -- the
    batch scheduler the reference's per-connection goroutine (main.go:12-21) lacks.  `on_frames(slot, frames)` receives
    the cropped I420 frames of a connection in decoding order; `on_close(slot, n_frames)` when its peer is done."""

    def __init__(self, max_connections=8, max_width=1920, max_height=1088, frames_per_batch=30, on_frames=None, on_close=None, read_size=1 << 16):
        self.decoder = Decoder(max_streams=max_connections, max_width=(max_width + 15) // 16 * 16, max_height=(max_height + 15) // 16 * 16,
                               max_frames_per_batch=frames_per_batch, max_slices_per_frame=16)
        self.decoder.set_isolation(True)  # one bad client must not take the other connections' chunks down with it
        self.errors = [0] * max_connections
        self.n = max_connections
        self.frames_per_batch = frames_per_batch
        self.on_frames, self.on_close, self.read_size = on_frames, on_close, read_size
        self.conn = [None] * self.n       # socket-like object per slot
        self.split = [None] * self.n
        self.queue = [[] for _ in range(self.n)]  # chunks waiting to be decoded
        self.eof = [False] * self.n
        self.count = [0] * self.n
        self.dims = [(0, 0)] * self.n

    def add(self, connection):
        """Attach a connection to a free slot; returns the slot or -1 when the server is full."""
        for i in range(self.n):
            if self.conn[i] is None:
                self.conn[i], self.split[i], self.queue[i], self.eof[i], self.count[i] = connection, AccessUnitSplitter(self.frames_per_batch), [], False, 0
                self.dims[i], self.errors[i] = (0, 0), 0
                self.decoder.reset_stream(i)  # nothing of the slot's previous client (parameter sets, reference pictures) survives
                if hasattr(connection, "setblocking"):
                    connection.setblocking(False)
                return i
        return -1

    def _pump(self, i):
        """Read what the connection has; returns True if anything arrived or it closed."""
        c = self.conn[i]
        try:
            data = c.recv(self.read_size) if hasattr(c, "recv") else c.read(self.read_size)
        except (BlockingIOError, InterruptedError):
            return False
        if data:
            self.queue[i] += self.split[i].feed(data)
        else:
            self.queue[i] += self.split[i].flush() + [END_OF_STREAM]  # (a lone first field at the end of the connection still goes out)
            self.eof[i] = True
        return True

    def tick(self):
        """One scheduling step: read every connection, decode one chunk per connection in one batch.  Returns the number
        of frames decoded."""
        for i in range(self.n):
            if self.conn[i] is not None and not self.eof[i]:
                self._pump(i)
        batch = [self.queue[i].pop(0) if (self.conn[i] is not None and self.queue[i]) else b"" for i in range(self.n)]
        total = 0
        if any(batch):
            self.decoder.decode(batch)
            for i in range(self.n):
                if not batch[i]:
                    continue
                st = self.decoder.stream_status(i)
                if st != 0:  # this connection's chunk was malformed / out of scope: close it, the others go on
                    self.errors[i] = st
                    self.queue[i] = []
                    self.eof[i] = True
                    continue
                k = self.decoder.frame_count(i)
                if not k:
                    continue
                frames = [self.decoder.read_frame_tight(i, f, crop=True) for f in range(k)]
                fi = self.decoder.frame_info(i, k - 1)
                self.dims[i] = (fi.width, fi.height)
                # frames of one geometry go out together (a chunk may span a resolution change)
                start = 0
                for f in range(1, k + 1):
                    if f == k or len(frames[f]) != len(frames[start]):
                        if self.on_frames:
                            self.on_frames(i, np.stack(frames[start:f]))
                        start = f
                self.count[i] += k
                total += k
        for i in range(self.n):
            if self.conn[i] is not None and self.eof[i] and not self.queue[i]:
                c, n = self.conn[i], self.count[i]
                self.conn[i] = None
                if hasattr(c, "close"):
                    c.close()
                if self.on_close:
                    self.on_close(i, n)
        return total

    def active(self):
        return sum(c is not None for c in self.conn)

    def run(self, idle_sleep=0.001):
        """Serve until every attached connection has closed."""
        import time
        while self.active():
            if not self.tick():
                time.sleep(idle_sleep)
        return list(self.count)


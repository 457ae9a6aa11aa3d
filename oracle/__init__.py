"""ctypes wrapper of the CPU oracle (oracle/).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package h264decode_amd/."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libh264oracle.so")


class StreamInfo(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("coded_width", ctypes.c_int),
                ("coded_height", ctypes.c_int), ("n_frames", ctypes.c_int), ("error", ctypes.c_int),
                ("n_mbs", ctypes.c_uint64), ("n_bins", ctypes.c_uint64), ("n_bits", ctypes.c_uint64)]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        L.h264o_decoder_create.restype = ctypes.c_void_p
        L.h264o_decoder_destroy.argtypes = [ctypes.c_void_p]
        L.h264o_last_error.restype = ctypes.c_char_p
        L.h264o_last_error.argtypes = [ctypes.c_void_p]
        L.h264o_decode_stream.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(StreamInfo)]
        L.h264o_set_mb_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.h264o_last_features.argtypes = [ctypes.c_void_p]
        L.h264o_last_features.restype = ctypes.c_uint32
        L.h264o_last_pocs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _lib = L
    return _lib


last_features = 0   # picture-management operations the last decode() executed (bit set, see h264o.h)
last_pocs = np.zeros(0, dtype=np.int32)  # PicOrderCnt of every picture the last decode() produced


class OracleError(RuntimeError):
    pass


def probe(stream: bytes) -> "StreamInfo":
    """Frame count and picture size of a stream (a full decode without output)."""
    L = lib()
    d = L.h264o_decoder_create()
    try:
        info = StreamInfo()
        r = L.h264o_decode_stream(d, stream, len(stream), 1, None, 0, ctypes.byref(info))
        if r < 0:
            raise OracleError(L.h264o_last_error(d).decode())
        return info
    finally:
        L.h264o_decoder_destroy(d)


def decode(stream: bytes, crop=True, trace=False, info=None):
    """Decode an Annex-B stream.  Returns (frames uint8[n, w*h*3/2], info[, trace int32[nmb,8]]).
    With `info` (from probe()) the sizing pass is skipped: exactly one decode runs (used for timing)."""
    L = lib()
    d = L.h264o_decoder_create()
    try:
        if info is None:
            info = StreamInfo()
            r = L.h264o_decode_stream(d, stream, len(stream), int(crop), None, 0, ctypes.byref(info))
            if r < 0:
                raise OracleError(L.h264o_last_error(d).decode())
            L.h264o_decoder_destroy(d)
            d = L.h264o_decoder_create()
        w, h = (info.width, info.height) if crop else (info.coded_width, info.coded_height)
        out = np.zeros((info.n_frames, w * h * 3 // 2), dtype=np.uint8)
        tr = None
        if trace:
            tr = np.zeros((int(info.n_mbs), 8), dtype=np.int32)
            L.h264o_set_mb_trace(d, tr.ctypes.data, tr.shape[0])
        r = L.h264o_decode_stream(d, stream, len(stream), int(crop), out.ctypes.data, out.nbytes, ctypes.byref(info))
        if r < 0:
            raise OracleError(L.h264o_last_error(d).decode())
        global last_features, last_pocs
        last_features = int(L.h264o_last_features(d))
        n = L.h264o_last_pocs(d, None, 0)
        last_pocs = np.zeros(n, dtype=np.int32)
        L.h264o_last_pocs(d, last_pocs.ctypes.data, n)
        return (out, info, tr) if trace else (out, info)
    finally:
        L.h264o_decoder_destroy(d)

"""ctypes wrapper of the synthetic H.264 stream generator (streamgen/).

Input synthesis only: produces Annex-B streams plus the generator's own closed-loop
reconstruction.  Not part of the decode product and not the oracle."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libstreamgen.so")


class Params(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "width", "height", "frames", "profile_idc", "cabac", "qp", "qp_jitter", "idr_period", "slices",
        "transform8x8", "num_ref_frames", "deblock_idc", "alpha_off_div2", "beta_off_div2", "cabac_init_idc",
        "constrained_intra", "chroma_qp_offset", "pcm_permille", "intra_in_p_permille", "skip_permille",
        "sub8x8_permille", "weighted_pred", "scaling_matrix", "noise")] + [("seed", ctypes.c_uint32),
        ("long_start_code", ctypes.c_int), ("poc_type", ctypes.c_int), ("rplm", ctypes.c_int), ("mmco", ctypes.c_int),
        ("idr_long_term", ctypes.c_int), ("nonref_period", ctypes.c_int), ("slice_qp_delta", ctypes.c_int), ("bframes", ctypes.c_int),
        ("direct_temporal", ctypes.c_int), ("weighted_bipred", ctypes.c_int), ("bskip_permille", ctypes.c_int),
        ("motion_x4", ctypes.c_int), ("motion_y4", ctypes.c_int), ("interlace_sps", ctypes.c_int), ("fn_gap_period", ctypes.c_int), ("fn_gap_declared", ctypes.c_int),
        ("b_pyramid", ctypes.c_int), ("slice_groups", ctypes.c_int), ("fmo_type", ctypes.c_int), ("aso", ctypes.c_int), ("field_pics", ctypes.c_int), ("poc_bottom_delta", ctypes.c_int), ("mono", ctypes.c_int)]


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
        _lib = ctypes.CDLL(_LIB)
        _lib.sg_encode.restype = ctypes.c_size_t
        _lib.sg_encode.argtypes = [ctypes.POINTER(Params), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                   ctypes.c_size_t, ctypes.c_void_p]
        _lib.sg_default_params.argtypes = [ctypes.POINTER(Params)]
        _lib.sg_last_error.restype = ctypes.c_char_p
        _lib.sg_source_frame.argtypes = [ctypes.POINTER(Params), ctypes.c_int, ctypes.c_void_p]
        _lib.sg_last_pocs.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _lib.sg_last_features.restype = ctypes.c_uint32
    return _lib


def default_params(**kw):
    p = Params()
    lib().sg_default_params(ctypes.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def encode(want_recon=True, **kw):
    """Returns (stream bytes, recon uint8[frames, coded_h*coded_w*3/2] or None, frame_sizes)."""
    p = default_params(**kw)
    W, H = (p.width + 15) & ~15, (p.height + 15) & ~15
    fsz = W * H * 3 // 2
    cap = max(1 << 20, p.frames * fsz * 2)
    stream = np.zeros(cap, dtype=np.uint8)
    recon = np.zeros((p.frames, fsz), dtype=np.uint8) if want_recon else None
    sizes = np.zeros(p.frames, dtype=np.uint32)
    n = lib().sg_encode(ctypes.byref(p), stream.ctypes.data, cap, recon.ctypes.data if want_recon else None,
                        recon.nbytes if want_recon else 0, sizes.ctypes.data)
    if n == 0:
        raise RuntimeError("streamgen failed: %s" % lib().sg_last_error().decode())
    return stream[:n].tobytes(), recon, sizes


def last_pocs():
    """PicOrderCnt the generator intended for every picture of the last encode() call."""
    n = lib().sg_last_pocs(None, 0)
    out = np.zeros(n, dtype=np.int32)
    lib().sg_last_pocs(out.ctypes.data, n)
    return out


def last_features():
    """Bit set of the picture-management syntax the last encode() call emitted (see sg.h)."""
    return int(lib().sg_last_features())


# Named recipes.  Sizes may be overridden for small test cases.
RECIPES = {
    "C1": dict(width=176, height=144, profile_idc=66, cabac=0, frames=30, idr_period=1, qp=28, seed=1),
    "C2": dict(width=1280, height=720, profile_idc=66, cabac=0, frames=60, idr_period=1, qp=28, seed=2),
    "C3": dict(width=1920, height=1080, profile_idc=77, cabac=1, frames=300, idr_period=30, qp=28, seed=3),
    "C4": dict(width=3840, height=2160, profile_idc=100, cabac=1, transform8x8=1, slices=8, frames=120,
               idr_period=30, qp=30, seed=4),
}


def recipe(name, **over):
    kw = dict(RECIPES[name])
    kw.update(over)
    return kw

"""ctypes binding of libh264mi.so (C ABI in include/h264mi.h).

The ctypes Structures are generated from the header itself so that they cannot drift from it.
There is no fallback: if the shared library is missing, load() raises."""
import ctypes
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_HEADER = os.path.join(_HERE, "..", "include", "h264mi.h")
_LIBPATH = os.environ.get("H264MI_LIB") or os.path.join(_HERE, "libh264mi.so")  # H264MI_LIB: a diagnostic build of the same library (csrc/Makefile)

_CT = {"int32_t": ctypes.c_int32, "uint32_t": ctypes.c_uint32, "int64_t": ctypes.c_int64, "uint8_t": ctypes.c_uint8,
       "double": ctypes.c_double, "void": None}


class H264MIError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("h264mi error %d: %s" % (code, msg))
        self.code = code


def _parse_structs():
    src = open(_HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef struct \{(.*?)\}\s*(h264mi_\w+);", src, flags=re.S):
        fields = []
        for stmt in body.split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            m = re.match(r"(void \*|\w+)\s*(.*)", stmt)
            typ, rest = m.group(1), m.group(2)
            for decl in rest.split(","):
                decl = decl.strip()
                dm = re.match(r"(\*?)(\w+)((?:\[\d+\])*)", decl)
                ptr, fname, dims = dm.group(1), dm.group(2), [int(x) for x in re.findall(r"\[(\d+)\]", dm.group(3))]
                ct = ctypes.c_void_p if (typ == "void *" or ptr) else _CT[typ]
                for n in reversed(dims):
                    ct = ct * n
                fields.append((fname, ct))
        out[name] = type(name, (ctypes.Structure,), {"_fields_": fields})
    return out


_S = _parse_structs()
Nal, Sps, Pps, SliceHdr, Config, BatchInfo, FrameInfo = (_S["h264mi_nal"], _S["h264mi_sps"], _S["h264mi_pps"], _S["h264mi_slice_header"],
                                                         _S["h264mi_config"], _S["h264mi_batch_info"], _S["h264mi_frame_info"])


def build(force=False):
    """Compile libh264mi.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", csrc])
    return _LIBPATH


_lib = None
_hooks = None


def load_hooks():
    """The hooks build of the same library (csrc/Makefile: mi_api.cpp with -DH264MI_TEST_HOOKS): the h264mi_internal_* entry points the tests and
    tools use.  They work on decoders of the product library too (plain structs, one HIP runtime per process); the product library itself exports
    the ABI of include/h264mi.h and nothing else."""
    global _hooks
    if _hooks is None:
        path = os.path.join(_HERE, "libh264mi_hooks.so")
        if not os.path.exists(path):
            raise H264MIError(-4, "libh264mi_hooks.so is not built (make -C h264decode_amd/csrc)")
        _hooks = ctypes.CDLL(path)
    return _hooks


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIBPATH):
        raise H264MIError(-4, "libh264mi.so is not built (run h264decode_amd.build() / __graft_entry__.build()); "
                              "there is no CPU fallback for the decode path")
    L = ctypes.CDLL(_LIBPATH)
    P, I32, SZ = ctypes.POINTER, ctypes.c_int32, ctypes.c_size_t
    u8p, vp = ctypes.c_char_p, ctypes.c_void_p
    sig = {
        "h264mi_annexb_scan": [u8p, SZ, P(Nal), I32, P(I32)],
        "h264mi_nal_parse": [u8p, SZ, P(Nal), vp, P(SZ)],
        "h264mi_sps_parse": [u8p, SZ, P(Sps)],
        "h264mi_pps_parse": [P(Sps), u8p, SZ, P(Pps)],
        "h264mi_slice_header_parse": [P(Sps), P(Pps), I32, I32, u8p, SZ, P(SliceHdr)],
        "h264mi_slice_starts_picture": [P(Sps), P(SliceHdr), P(SliceHdr)],
        "h264mi_pps_slice_group_ids": [P(Sps), u8p, SZ, vp, SZ, P(SZ)],
        "h264mi_map_unit_to_slice_group_map": [P(Sps), P(Pps), vp, SZ, I32, vp, SZ, P(SZ)],
        "h264mi_mb_to_slice_group_map": [P(Sps), P(Pps), vp, SZ, I32, I32, vp, SZ, P(SZ)],
        "h264mi_next_mb_address": [vp, SZ, SZ],
        "h264mi_init": [I32],
        "h264mi_decoder_create": [P(Config), P(vp)],
        "h264mi_decoder_destroy": [vp],
        "h264mi_decoder_set_stream": [vp, vp],
        "h264mi_decoder_reset": [vp],
        "h264mi_stream_reset": [vp, I32],
        "h264mi_stream_status": [vp, I32, P(I32)],
        "h264mi_decoder_set_isolation": [vp, I32],
        "h264mi_frame_get_info": [vp, I32, I32, P(FrameInfo)],
        "h264mi_stream_output_order": [vp, I32, P(I32), I32, P(I32)],
        "h264mi_batch_prepare": [vp, I32, P(vp), P(SZ), P(BatchInfo)],
        "h264mi_batch_execute": [vp],
        "h264mi_batch_sync": [vp],
        "h264mi_decode_batch": [vp, I32, P(vp), P(SZ), P(BatchInfo)],
        "h264mi_stream_frame_count": [vp, I32, P(I32)],
        "h264mi_frame_device_planes": [vp, I32, I32, P(vp), P(vp), P(vp), P(I32), P(I32), P(I32), P(I32)],
        "h264mi_frame_read": [vp, I32, I32, I32, vp, SZ],
        "h264mi_frame_pack_device": [vp, I32, I32, vp, SZ],
        "h264mi_batch_pack_device": [vp, I32, vp, SZ, P(SZ)],
        "h264mi_frame_read_mbrecs": [vp, I32, I32, vp, SZ],
        "h264mi_frame_read_mbmv1": [vp, I32, I32, vp, SZ],
        "h264mi_decoder_set_profiling": [vp, I32],
        "h264mi_decoder_memory": [vp, P(ctypes.c_int64)],
        "h264mi_decoder_coef_pool": [vp, P(ctypes.c_int64), P(ctypes.c_int64)],
        "h264mi_decoder_unpinned_failures": [vp, P(ctypes.c_int64)],
        "h264mi_last_kernel_times": [vp, P(ctypes.c_double)],
        "h264mi_last_launch_times": [vp, I32, P(ctypes.c_float), I32, P(I32)],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = I32
    L.h264mi_last_error_string.restype = ctypes.c_char_p
    L.h264mi_version.restype = ctypes.c_char_p
    _lib = L
    return L


def lib():
    return load()


def check(code):
    if code != 0:
        raise H264MIError(code, load().h264mi_last_error_string().decode(errors="replace"))


EXPORTS = ["h264mi_annexb_scan", "h264mi_nal_parse", "h264mi_sps_parse", "h264mi_pps_parse", "h264mi_slice_header_parse",
           "h264mi_init", "h264mi_decoder_create", "h264mi_decoder_destroy", "h264mi_decoder_set_stream", "h264mi_decoder_reset",
           "h264mi_batch_prepare", "h264mi_batch_execute", "h264mi_batch_sync", "h264mi_decode_batch", "h264mi_stream_frame_count",
           "h264mi_frame_device_planes", "h264mi_frame_read", "h264mi_frame_pack_device", "h264mi_frame_read_mbrecs",
           "h264mi_decoder_set_profiling", "h264mi_last_kernel_times", "h264mi_last_error_string", "h264mi_version",
           "h264mi_last_launch_times", "h264mi_batch_pack_device", "h264mi_stream_reset", "h264mi_stream_status", "h264mi_decoder_set_isolation", "h264mi_frame_get_info", "h264mi_stream_output_order", "h264mi_decoder_memory", "h264mi_frame_read_mbmv1", "h264mi_decoder_coef_pool", "h264mi_decoder_unpinned_failures",
           "h264mi_slice_starts_picture", "h264mi_pps_slice_group_ids", "h264mi_map_unit_to_slice_group_map", "h264mi_mb_to_slice_group_map", "h264mi_next_mb_address"]

"""medical-image-codec_amd: MI355X (gfx950) implementation of MIC's parallel-strip hot path.

This package is a thin ctypes binding over the C ABI of ``libmic_hip.so``
(``include/mic_hip.h``) -- the same entry points the reference's Go package binds through
cgo.  Function names mirror the reference's Go API
(``parallelstrips.go``, ``multiframecompress.go``) so tests read like the reference's.

There is no CPU fallback: importing works without a GPU (for symbol checks), but every
codec call raises ``MicError`` unless the HIP library runs on a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIC_HIP_LIB") or os.path.join(_HERE, "libmic_hip.so")   # (MIC_HIP_LIB: an A/B build of the same ABI, tools/ only)

MIC_OK = 0
MIC_ERR_ARGS = -1
MIC_ERR_NOMEM = -2
MIC_ERR_USE_RLE = -3
MIC_ERR_CAPACITY = -5
MIC_ERR_CORRUPT = -6
MIC_ERR_DEVICE = -7
MIC_ERR_INTERNAL = -8
MIC_ERR_UNSUPPORTED = -9
MIC_HIP_PRED_GRAD = 0x200          # OR'ed into a session unit's nstates: gradient-adaptive predictor (include/mic_hip.h)
MIC_ERR_INCOMPRESSIBLE = -10

_ERR_NAMES = {
    MIC_ERR_ARGS: "bad arguments", MIC_ERR_NOMEM: "out of memory",
    MIC_ERR_USE_RLE: "input is single value repeated",        # ErrUseRLE, fseu16.go:36
    MIC_ERR_CAPACITY: "output buffer too small", MIC_ERR_CORRUPT: "corrupt stream",
    MIC_ERR_DEVICE: "no usable gfx950 device / HIP error", MIC_ERR_INTERNAL: "internal error",
    MIC_ERR_UNSUPPORTED: "unsupported", MIC_ERR_INCOMPRESSIBLE: "input is not compressible",  # fseu16.go:33
}


class MicError(RuntimeError):
    def __init__(self, code: int, where: str = ""):
        self.code = code
        super().__init__(f"{where}: {_ERR_NAMES.get(code, 'error')} (rc={code})")


class ErrUseRLE(MicError):
    """Reference sentinel ErrUseRLE (fseu16.go:36)."""


class ErrIncompressible(MicError):
    """Reference sentinel ErrIncompressible (fseu16.go:33)."""


def _raise(code: int, where: str):
    if code == MIC_ERR_USE_RLE:
        raise ErrUseRLE(code, where)
    if code == MIC_ERR_INCOMPRESSIBLE:
        raise ErrIncompressible(code, where)
    raise MicError(code, where)


class EncJob(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("max_value", C.c_uint16), ("nstates", C.c_uint16),
                ("out", C.c_void_p), ("out_cap", C.c_size_t), ("out_len", C.c_size_t),
                ("status", C.c_int32), ("nstates_used", C.c_int32)]


class DecJob(C.Structure):
    _fields_ = [("compressed", C.c_void_p), ("compressed_len", C.c_size_t),
                ("pixels_out", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("status", C.c_int32)]


class PicsEncJob(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("max_value", C.c_uint16), ("nstates", C.c_uint16), ("num_strips", C.c_int32),
                ("out", C.c_void_p), ("out_cap", C.c_size_t), ("out_len", C.c_size_t), ("status", C.c_int32),
                ("failed_strip", C.c_int32)]


class PicsDecJob(C.Structure):
    _fields_ = [("compressed", C.c_void_p), ("compressed_len", C.c_size_t),
                ("pixels_out", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("status", C.c_int32),
                ("failed_strip", C.c_int32)]


class Unit(C.Structure):
    _fields_ = [("px_offset", C.c_uint64), ("width", C.c_int32), ("height", C.c_int32),
                ("max_value", C.c_uint16), ("nstates", C.c_uint16)]


# every symbol include/mic_hip.h declares (tests/test_abi.py checks the .so exports them all)
ABI_SYMBOLS = [
    "mic_hip_set_device", "mic_hip_set_devices", "mic_hip_get_devices", "mic_hip_shard_plan", "mic_hip_device_name", "mic_hip_version",
    "mic_hip_compress_frame", "mic_hip_decompress_frame",
    "mic_hip_fse_compress_u16", "mic_hip_fse_decompress_u16_auto", "mic_hip_fse_compress_u16_ex", "mic_hip_fse_decompress_u16_ex",
    "mic_hip_compress_batch", "mic_hip_decompress_batch", "mic_hip_host_alloc", "mic_hip_host_free",
    "mic_hip_pics_compress", "mic_hip_pics_compress_ex", "mic_hip_pics_info", "mic_hip_pics_decompress", "mic_hip_pics_decompress_ex",
    "mic_hip_pics_compress_batch", "mic_hip_pics_decompress_batch",
    "mic_hip_mic2_compress", "mic_hip_mic2_compress_temporal", "mic_hip_mic2_info", "mic_hip_mic2_decompress",
    "mic_hip_mic2_decompress_frame",
    "mic_hip_wavelet_v2_compress", "mic_hip_wavelet_v2_compress_batch", "mic_hip_wavelet_v2_decompress_batch", "mic_hip_wavelet_v2_info", "mic_hip_wavelet_v2_decompress",
    "mic_hip_compress_frame_grad", "mic_hip_decompress_frame_grad", "mic_hip_pica_compress", "mic_hip_pica_info", "mic_hip_pica_decompress",
    "mic_hip_rgb_compress", "mic_hip_rgb_decompress", "mic_hip_micr_compress", "mic_hip_micr_info", "mic_hip_micr_decompress",
    "mic_hip_mic1_compress", "mic_hip_mic1_info", "mic_hip_mic1_decompress",
    "mic_hip_wsi_compress", "mic_hip_wsi_compress_ex", "mic_hip_wsi_format", "mic_hip_wsi_info", "mic_hip_wsi_level_info",
    "mic_hip_wsi_decompress_tile", "mic_hip_wsi_decompress_level", "mic_hip_wsi_decompress_region",
    "mic_hip_session_create", "mic_hip_session_create_on", "mic_hip_session_device", "mic_hip_session_workspace_bytes", "mic_hip_session_destroy", "mic_hip_session_stream",
    "mic_hip_device_copy",
    "mic_hip_session_wavelet_v2_encode", "mic_hip_session_wavelet_v2_decode",
    "mic_hip_session_wsi_encode", "mic_hip_session_wsi_write", "mic_hip_session_wsi_payload", "mic_hip_session_wsi_decode_level", "mic_hip_session_wsi_levels",
    "mic_hip_session_encode", "mic_hip_session_decode",
    "mic_hip_session_encode_enqueue", "mic_hip_session_encode_finish",
    "mic_hip_session_decode_enqueue", "mic_hip_session_decode_finish",
    "mic_hip_session_set_timing", "mic_hip_session_last_timings",
]

_lib: Optional[C.CDLL] = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels carry their own libamdhip64.so (SONAME libamdhip64.so.7) and ask for it by
    the bare name, which the loader does not match against /opt/rocm's copy once libmic_hip.so has pulled that in: a process that
    loads this library first and imports torch afterwards ends up with two runtimes, and the second one finds no GPU.  When a torch
    install is present its copy is loaded first (by path, without importing torch), so either import order gives one runtime.
    MIC_HIP_NO_TORCH_PRELOAD=1 turns the preload off (a process that never imports torch, or whose torch wheel is built against another
    HIP major version than libmic_hip.so: binding this library to that runtime would be an ABI mismatch nobody reports)."""
    import importlib.util
    if "torch" in sys.modules or os.environ.get("MIC_HIP_NO_TORCH_PRELOAD") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError as e:
            import warnings
            warnings.warn(f"mic_hip: could not preload torch's HIP runtime ({cand}: {e}); importing torch after this library may "
                          "leave the process with two runtimes", RuntimeWarning)


def _check_one_hip_runtime():
    """After libmic_hip.so is mapped: exactly one libamdhip64 in the process, or say so loudly (two runtimes = the second finds no
    GPU; see _share_torch_hip_runtime)."""
    try:
        with open("/proc/self/maps") as f:
            paths = {ln.split()[-1] for ln in f if "libamdhip64" in ln}
    except OSError:
        return
    real = {os.path.realpath(q) for q in paths}
    if len(real) > 1:
        import warnings
        warnings.warn("mic_hip: more than one HIP runtime is mapped into this process: " + ", ".join(sorted(real)) +
                      " -- device calls of the one loaded second will fail; import torch before this package, or set "
                      "MIC_HIP_NO_TORCH_PRELOAD=1 and never import torch", RuntimeWarning)


def lib() -> C.CDLL:
    """Loads libmic_hip.so (built in-tree by csrc/build.sh); fails loudly when missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run medical-image-codec_amd/csrc/build.sh "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    _check_one_hip_runtime()
    L.mic_hip_device_name.restype = C.c_char_p
    L.mic_hip_version.restype = C.c_char_p
    L.mic_hip_session_stream.restype = C.c_void_p
    L.mic_hip_session_stream.argtypes = [C.c_void_p]
    L.mic_hip_session_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_size_t]
    L.mic_hip_device_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.mic_hip_session_create_on.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_size_t]
    L.mic_hip_session_device.argtypes = [C.c_void_p]
    L.mic_hip_session_workspace_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.mic_hip_session_workspace_bytes.restype = C.c_size_t
    L.mic_hip_session_wavelet_v2_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                                    C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_int)]
    L.mic_hip_session_wavelet_v2_decode.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_int, C.c_int,
                                                    C.c_void_p, C.POINTER(C.c_int32)]
    L.mic_hip_session_wsi_encode.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mic_hip_session_wsi_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_session_wsi_payload.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p, C.c_size_t]
    L.mic_hip_session_wsi_decode_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.mic_hip_session_wsi_levels.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.mic_hip_session_destroy.argtypes = [C.c_void_p]
    L.mic_hip_session_destroy.restype = None
    L.mic_hip_compress_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint16, C.c_int,
                                         C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_decompress_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
    L.mic_hip_fse_compress_u16.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_fse_decompress_u16_auto.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_fse_compress_u16_ex.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_fse_decompress_u16_ex.argtypes = [C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_compress_batch.argtypes = [C.POINTER(EncJob), C.c_int]
    L.mic_hip_decompress_batch.argtypes = [C.POINTER(DecJob), C.c_int]
    L.mic_hip_pics_compress.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint16, C.c_int, C.c_int,
                                        C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_pics_compress_ex.argtypes = L.mic_hip_pics_compress.argtypes + [C.POINTER(C.c_int)]
    L.mic_hip_pics_decompress_ex.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.mic_hip_set_devices.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.mic_hip_get_devices.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.mic_hip_shard_plan.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.mic_hip_pics_compress_batch.argtypes = [C.POINTER(PicsEncJob), C.c_int]
    L.mic_hip_pics_decompress_batch.argtypes = [C.POINTER(PicsDecJob), C.c_int]
    L.mic_hip_host_alloc.argtypes = [C.c_size_t]
    L.mic_hip_host_alloc.restype = C.c_void_p
    L.mic_hip_host_free.argtypes = [C.c_void_p]
    L.mic_hip_host_free.restype = None
    L.mic_hip_pics_info.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int)] * 4
    L.mic_hip_pics_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
    L.mic_hip_mic2_compress.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint16,
                                        C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_mic2_compress_temporal.argtypes = L.mic_hip_mic2_compress.argtypes
    L.mic_hip_mic2_info.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int)] * 4
    L.mic_hip_mic2_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.mic_hip_mic2_decompress_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
    L.mic_hip_wavelet_v2_compress.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint16, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_wavelet_v2_compress_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint16, C.c_int, C.c_void_p, C.c_size_t,
                                                    C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]
    L.mic_hip_wavelet_v2_decompress_batch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32)]
    L.mic_hip_wavelet_v2_info.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int)] * 4
    L.mic_hip_wavelet_v2_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.mic_hip_wsi_compress.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_compress_frame_grad.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint16, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_decompress_frame_grad.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
    L.mic_hip_pica_compress.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint16, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_pica_info.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int)] * 3
    L.mic_hip_pica_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
    L.mic_hip_rgb_compress.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_rgb_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.mic_hip_micr_compress.argtypes = L.mic_hip_rgb_compress.argtypes
    L.mic_hip_micr_info.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mic_hip_micr_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.mic_hip_mic1_compress.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint16, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_mic1_info.argtypes = L.mic_hip_micr_info.argtypes
    L.mic_hip_mic1_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.mic_hip_wsi_compress_ex.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.mic_hip_wsi_format.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int)] * 3
    L.mic_hip_wsi_info.argtypes = [C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int)] * 5 + [C.POINTER(C.c_uint64)]
    L.mic_hip_wsi_level_info.argtypes = [C.c_void_p, C.c_size_t, C.c_int] + [C.POINTER(C.c_int)] * 4
    L.mic_hip_wsi_decompress_tile.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mic_hip_wsi_decompress_level.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
    L.mic_hip_wsi_decompress_region.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mic_hip_session_encode_enqueue.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Unit), C.c_int]
    L.mic_hip_session_encode_finish.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                                C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.mic_hip_session_decode_enqueue.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64),
                                                 C.POINTER(Unit), C.c_int, C.c_void_p]
    L.mic_hip_session_decode_finish.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.mic_hip_session_set_timing.argtypes = [C.c_void_p, C.c_int]
    L.mic_hip_session_last_timings.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    _lib = L
    return L


def device_name() -> str:
    return lib().mic_hip_device_name().decode()


def device_copy(d_dst: int, d_src: int, nbytes: int) -> None:
    """device -> device copy (a session's result buffers are reused by its next call)"""
    rc = lib().mic_hip_device_copy(d_dst, d_src, nbytes)
    if rc:
        _raise(rc, "device_copy")


def _frame_bound(npx: int) -> int:
    """MIC_HIP_FRAME_BOUND (include/mic_hip.h): worst-case bytes of one coded frame / strip"""
    return 4 * npx + 135168


def _u16(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint16)
    return a


def _bytes_arr(b) -> np.ndarray:
    return np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b, dtype=np.uint8)


# ------------------------------------------------------------------ bare FSE stage
def fse_compress_u16(symbols, flavour: int = 2, table_log: int = 0) -> bytes:
    """FSECompressU16 / TwoState / FourState / EightState (flavour 1/2/4/8) and
    RANSCompressU16EightState (flavour 108); table_log = ScratchU16.TableLog (fseu16.go:101-102; 0 = default)."""
    sym = _u16(symbols).reshape(-1)
    cap = sym.size * 2 + 200000
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().mic_hip_fse_compress_u16_ex(sym.ctypes.data, sym.size, flavour, table_log, out.ctypes.data, cap, C.byref(n))
    if rc:
        _raise(rc, "fse_compress_u16")
    return out[: n.value].tobytes()


def fse_decompress_u16_auto(data, cap: int, decompress_limit: int = 0) -> np.ndarray:
    """FSEDecompressU16Auto (fse2state.go:102-116); decompress_limit = ScratchU16.DecompressLimit (fseu16.go:87-91; 0 = default)."""
    c = _bytes_arr(data)
    out = np.empty(cap, dtype=np.uint16)
    n = C.c_size_t(0)
    rc = lib().mic_hip_fse_decompress_u16_ex(c.ctypes.data, c.size, decompress_limit, out.ctypes.data, cap, C.byref(n))
    if rc:
        _raise(rc, "fse_decompress_u16_auto")
    return out[: n.value].copy()


# ------------------------------------------------------------------ unit codec
def compress_single_frame(pixels, width: int, height: int, max_value: int, nstates: int = 2) -> bytes:
    """CompressSingleFrame / 4State / 8State (multiframecompress.go:15,38,67)."""
    px = _u16(pixels).reshape(-1)
    if px.size != width * height:
        raise MicError(MIC_ERR_ARGS, "compress_single_frame")
    cap = _frame_bound(px.size)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().mic_hip_compress_frame(px.ctypes.data, width, height, max_value, nstates, out.ctypes.data, cap, C.byref(n))
    if rc:
        _raise(rc, "compress_single_frame")
    return out[: n.value].tobytes()


def decompress_single_frame(compressed, width: int, height: int) -> np.ndarray:
    """DecompressSingleFrame (multiframecompress.go:97)."""
    c = _bytes_arr(compressed)
    out = np.empty(width * height, dtype=np.uint16)
    rc = lib().mic_hip_decompress_frame(c.ctypes.data, c.size, out.ctypes.data, width, height)
    if rc:
        _raise(rc, "decompress_single_frame")
    return out.reshape(height, width)


def compress_batch(frames: Sequence[np.ndarray], max_values: Sequence[int], nstates: int = 2
                   ) -> List[Tuple[int, bytes, int]]:
    """One launch chain over many frames; returns [(status, blob, nstates_used)]."""
    n = len(frames)
    arrs = [_u16(f) for f in frames]
    outs = [np.empty(_frame_bound(a.size), dtype=np.uint8) for a in arrs]
    jobs = (EncJob * n)()
    for i, a in enumerate(arrs):
        h, w = a.shape
        jobs[i].pixels = a.ctypes.data; jobs[i].width = w; jobs[i].height = h
        jobs[i].max_value = int(max_values[i]); jobs[i].nstates = nstates
        jobs[i].out = outs[i].ctypes.data; jobs[i].out_cap = outs[i].size
    rc = lib().mic_hip_compress_batch(jobs, n)
    if rc:
        _raise(rc, "compress_batch")
    return [(jobs[i].status, outs[i][: jobs[i].out_len].tobytes() if jobs[i].status == 0 else b"", jobs[i].nstates_used)
            for i in range(n)]


def decompress_batch(blobs: Sequence[bytes], dims: Sequence[Tuple[int, int]]) -> List[Tuple[int, Optional[np.ndarray]]]:
    n = len(blobs)
    cs = [_bytes_arr(b) for b in blobs]
    outs = [np.empty(w * h, dtype=np.uint16) for (w, h) in dims]
    jobs = (DecJob * n)()
    for i in range(n):
        jobs[i].compressed = cs[i].ctypes.data; jobs[i].compressed_len = cs[i].size
        jobs[i].pixels_out = outs[i].ctypes.data; jobs[i].width = dims[i][0]; jobs[i].height = dims[i][1]
    rc = lib().mic_hip_decompress_batch(jobs, n)
    if rc:
        _raise(rc, "decompress_batch")
    return [(jobs[i].status, outs[i].reshape(dims[i][1], dims[i][0]) if jobs[i].status == 0 else None) for i in range(n)]


# ------------------------------------------------------------------ PICS
def compress_parallel_strips(pixels, width: int, height: int, max_value: int, num_strips: int, nstates: int = 2) -> bytes:
    """CompressParallelStrips / 4State / 8State (parallelstrips.go:55,128,199); a strip's error carries its index, as the
    reference's "parallelstrips: strip %d: %w" does (:97): MicError.strip."""
    px = _u16(pixels).reshape(-1)
    if px.size != width * height:
        raise MicError(MIC_ERR_ARGS, "parallelstrips: pixel count != width*height")
    cap = px.size * 4 + 135168 * max(num_strips, 1) + 8 * max(num_strips, 1) + 20
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0); bad = C.c_int(-1)
    rc = lib().mic_hip_pics_compress_ex(px.ctypes.data, width, height, max_value, num_strips, nstates, out.ctypes.data, cap, C.byref(n), C.byref(bad))
    if rc:
        try:
            _raise(rc, "parallelstrips" + (f": strip {bad.value}" if bad.value >= 0 else ""))
        except MicError as e:
            e.strip = bad.value
            raise
    return out[: n.value].tobytes()


def set_devices(devices: Sequence[int]) -> None:
    """mic_hip_set_devices: the GPUs the batch entry points spread their jobs over (the first is the default device)."""
    arr = (C.c_int * len(devices))(*devices)
    rc = lib().mic_hip_set_devices(arr, len(devices))
    if rc:
        _raise(rc, "set_devices")


def get_devices() -> List[int]:
    arr = (C.c_int * 64)()
    n = lib().mic_hip_get_devices(arr, 64)
    return [arr[i] for i in range(min(n, 64))]


def shard_plan(weights: Sequence[int], shards: int) -> List[int]:
    """mic_hip_shard_plan: first[0..shards] of the contiguous cut the batch entry points make over several devices."""
    w = (C.c_uint64 * len(weights))(*[int(x) for x in weights])
    first = (C.c_int * (shards + 1))()
    rc = lib().mic_hip_shard_plan(w, len(weights), shards, first)
    if rc:
        _raise(rc, "shard_plan")
    return list(first)


def pics_bound(width: int, height: int, num_strips: int) -> int:
    """MIC_HIP_PICS_BOUND"""
    ns = max(num_strips, 1)
    return 20 + 8 * ns + 4 * width * height + 135168 * ns


def compress_parallel_strips_batch(images: Sequence[np.ndarray], max_value: int, num_strips: int, nstates: int = 2,
                                   outs: Optional[Sequence[np.ndarray]] = None) -> List[Tuple[int, "np.ndarray"]]:
    """Many images, one call (mic_hip_pics_compress_batch): [(status, file bytes as a uint8 view of its out buffer)].
    images: (height, width) uint16 arrays (ordinary or pinned memory); outs: caller buffers of >= pics_bound bytes, or None."""
    n = len(images)
    arrs = [_u16(a) for a in images]
    if outs is None:
        outs = [np.empty(pics_bound(a.shape[1], a.shape[0], num_strips), dtype=np.uint8) for a in arrs]
    jobs = (PicsEncJob * n)()
    for i, a in enumerate(arrs):
        jobs[i].pixels = a.ctypes.data; jobs[i].width = a.shape[1]; jobs[i].height = a.shape[0]
        jobs[i].max_value = max_value; jobs[i].nstates = nstates; jobs[i].num_strips = num_strips
        jobs[i].out = outs[i].ctypes.data; jobs[i].out_cap = outs[i].size
    rc = lib().mic_hip_pics_compress_batch(jobs, n)
    if rc:
        _raise(rc, "compress_parallel_strips_batch")
    compress_parallel_strips_batch.failed_strips = [jobs[i].failed_strip for i in range(n)]     # (of the last call: index of each job's failing strip, -1)
    return [(jobs[i].status, outs[i][: jobs[i].out_len]) for i in range(n)]


def decompress_parallel_strips_batch(files: Sequence, dims: Sequence[Tuple[int, int]],
                                     outs: Optional[Sequence[np.ndarray]] = None) -> List[Tuple[int, "np.ndarray"]]:
    """Many PICS files, one call (mic_hip_pics_decompress_batch): [(status, (height, width) uint16 pixels)]."""
    n = len(files)
    cs = [_bytes_arr(f) for f in files]
    if outs is None:
        outs = [np.empty(w * h, dtype=np.uint16) for (w, h) in dims]
    jobs = (PicsDecJob * n)()
    for i in range(n):
        jobs[i].compressed = cs[i].ctypes.data; jobs[i].compressed_len = cs[i].size
        jobs[i].pixels_out = outs[i].ctypes.data; jobs[i].width = dims[i][0]; jobs[i].height = dims[i][1]
    rc = lib().mic_hip_pics_decompress_batch(jobs, n)
    if rc:
        _raise(rc, "decompress_parallel_strips_batch")
    return [(jobs[i].status, outs[i].reshape(dims[i][1], dims[i][0])) for i in range(n)]


def host_alloc(nbytes: int, dtype=np.uint8) -> np.ndarray:
    """A pinned host buffer (mic_hip_host_alloc) as a numpy array; free it with host_free(arr)."""
    p = lib().mic_hip_host_alloc(nbytes)
    if not p:
        raise MicError(MIC_ERR_NOMEM, "host_alloc")
    buf = (C.c_uint8 * nbytes).from_address(p)
    a = np.frombuffer(buf, dtype=np.uint8).view(dtype)
    _PINNED[p] = nbytes
    return a


def host_free(a: np.ndarray) -> None:
    """Frees the pinned allocation `a` lies in -- `a` itself, or any view of it (a reshape, a slice: the allocation is found by
    address).  A buffer that host_alloc did not hand out raises instead of leaking quietly."""
    addr = int(a.ctypes.data)
    for p, n in _PINNED.items():
        if p <= addr < p + max(n, 1):
            del _PINNED[p]
            lib().mic_hip_host_free(p)
            return
    raise ValueError("host_free: not (a view of) a buffer from host_alloc, or freed already")


_PINNED = {}


def pics_info(compressed) -> Tuple[int, int, int, int]:
    c = _bytes_arr(compressed)
    w, h, n, sh = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib().mic_hip_pics_info(c.ctypes.data, c.size, C.byref(w), C.byref(h), C.byref(n), C.byref(sh))
    if rc:
        _raise(rc, "parallelstrips")
    return w.value, h.value, n.value, sh.value


def decompress_parallel_strips(compressed) -> Tuple[np.ndarray, int, int]:
    """DecompressParallelStrips (parallelstrips.go:270): returns (pixels, width, height)."""
    c = _bytes_arr(compressed)
    w, h, _, _ = pics_info(c)
    out = np.empty(w * h, dtype=np.uint16)
    bad = C.c_int(-1)
    rc = lib().mic_hip_pics_decompress_ex(c.ctypes.data, c.size, out.ctypes.data, w, h, C.byref(bad))
    if rc:
        try:
            _raise(rc, "parallelstrips" + (f": strip {bad.value}" if bad.value >= 0 else ""))
        except MicError as e:
            e.strip = bad.value
            raise
    return out.reshape(h, w), w, h


# ------------------------------------------------------------------ MIC2
def compress_multi_frame(frames: np.ndarray, width: int, height: int, max_value: int, temporal: bool = False) -> bytes:
    """CompressMultiFrame (multiframecompress.go:179): independent frames, or the temporal pipeline
    (frame 0 spatial, ZigZag residuals of consecutive frames after it)."""
    fr = _u16(frames)
    nframes = fr.shape[0]
    cap = fr.size * 4 + 135168 * nframes + 8 * nframes + 20
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    fn = lib().mic_hip_mic2_compress_temporal if temporal else lib().mic_hip_mic2_compress
    rc = fn(fr.ctypes.data, width, height, nframes, max_value, out.ctypes.data, cap, C.byref(n))
    if rc:
        _raise(rc, "compress_multi_frame")
    return out[: n.value].tobytes()


def decompress_multi_frame(compressed) -> np.ndarray:
    """DecompressMultiFrame (multiframecompress.go:227)."""
    c = _bytes_arr(compressed)
    w, h, n, t = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib().mic_hip_mic2_info(c.ctypes.data, c.size, C.byref(w), C.byref(h), C.byref(n), C.byref(t))
    if rc:
        _raise(rc, "decompress_multi_frame")
    out = np.empty(max(n.value, 0) * max(w.value, 0) * max(h.value, 0), dtype=np.uint16)
    rc = lib().mic_hip_mic2_decompress(c.ctypes.data, c.size, out.ctypes.data, out.size)
    if rc:
        _raise(rc, "decompress_multi_frame")
    return out.reshape(n.value, h.value, w.value)


def decompress_frame(compressed, frame_idx: int) -> np.ndarray:
    """DecompressFrame (multiframecompress.go:266): one frame of a MIC2 file."""
    c = _bytes_arr(compressed)
    w, h, n, t = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib().mic_hip_mic2_info(c.ctypes.data, c.size, C.byref(w), C.byref(h), C.byref(n), C.byref(t))
    if rc:
        _raise(rc, "decompress_frame")
    out = np.empty(max(w.value, 0) * max(h.value, 0), dtype=np.uint16)
    rc = lib().mic_hip_mic2_decompress_frame(c.ctypes.data, c.size, frame_idx, out.ctypes.data, out.size)
    if rc:
        _raise(rc, "decompress_frame")
    return out.reshape(h.value, w.value)


# ------------------------------------------------------------------ WaveletV2
def wavelet_v2_compress(pixels, rows: int, cols: int, max_value: int, levels: int = 5) -> bytes:
    """WaveletV2RLEFSECompressU16 / WaveletV2SIMDRLEFSECompressU16 (waveletfsecompressu16.go:303, :374)."""
    px = _u16(pixels).reshape(-1)
    if px.size != rows * cols:
        raise MicError(MIC_ERR_ARGS, "pixel count does not match rows*cols")
    cap = px.size * 6 + 200000
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().mic_hip_wavelet_v2_compress(px.ctypes.data, rows, cols, max_value, levels, out.ctypes.data, cap, C.byref(n))
    if rc:
        _raise(rc, "wavelet_v2_compress")
    return out[: n.value].tobytes()


def wavelet_v2_compress_batch(frames, max_value: int, levels: int = 5) -> List[Tuple[int, bytes]]:
    """nframes x rows x cols frames of one shape in one launch chain: [(status, file)], each file as the single call writes it."""
    fr = np.ascontiguousarray(frames, dtype=np.uint16)
    nf, rows, cols = fr.shape
    stride = rows * cols * 6 + 200000
    out = np.empty(nf * stride, dtype=np.uint8)
    lens = (C.c_size_t * nf)(); st = (C.c_int32 * nf)()
    rc = lib().mic_hip_wavelet_v2_compress_batch(fr.ctypes.data, nf, rows, cols, max_value, levels, out.ctypes.data, stride, lens, st)
    if rc:
        _raise(rc, "wavelet_v2_compress_batch")
    return [(int(st[i]), out[i * stride: i * stride + lens[i]].tobytes() if st[i] == 0 else b"") for i in range(nf)]


def wavelet_v2_decompress_batch(files: Sequence[bytes]) -> Tuple[List[int], np.ndarray]:
    """files of ONE shape -> ([status], nframes x rows x cols uint16)."""
    cs = [_bytes_arr(b) for b in files]
    nf = len(cs)
    r, cc, mv, lv = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib().mic_hip_wavelet_v2_info(cs[0].ctypes.data, cs[0].size, C.byref(r), C.byref(cc), C.byref(mv), C.byref(lv))
    if rc:
        _raise(rc, "wavelet_v2_decompress_batch")
    out = np.zeros((nf, max(r.value, 0), max(cc.value, 0)), dtype=np.uint16)
    ptrs = (C.c_void_p * nf)(*[c.ctypes.data for c in cs]); lens = (C.c_size_t * nf)(*[c.size for c in cs]); st = (C.c_int32 * nf)()
    rc = lib().mic_hip_wavelet_v2_decompress_batch(ptrs, lens, nf, out.ctypes.data, out.size, st)
    if rc:
        _raise(rc, "wavelet_v2_decompress_batch")
    return [int(v) for v in st], out


def wavelet_v2_decompress(compressed) -> Tuple[np.ndarray, int, int]:
    """WaveletV2{,SIMD}RLEFSEDecompressU16 (:380, :493): returns (pixels, rows, cols)."""
    c = _bytes_arr(compressed)
    r, cc, mv, lv = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = lib().mic_hip_wavelet_v2_info(c.ctypes.data, c.size, C.byref(r), C.byref(cc), C.byref(mv), C.byref(lv))
    if rc:
        _raise(rc, "wavelet_v2_decompress")
    out = np.empty(max(r.value, 0) * max(cc.value, 0), dtype=np.uint16)
    rc = lib().mic_hip_wavelet_v2_decompress(c.ctypes.data, c.size, out.ctypes.data, out.size)
    if rc:
        _raise(rc, "wavelet_v2_decompress")
    return out.reshape(r.value, cc.value), r.value, cc.value


# ------------------------------------------------------------------ MIC3 / WSI
def compress_wsi(pixels, width: int, height: int, channels: int = 3, bits_per_sample: int = 8,
                 tile_w: int = 0, tile_h: int = 0, levels: int = 0) -> bytes:
    """CompressWSI (wsicompress.go:27): 8-bit RGB, or greyscale (channels=1) with 8 or 16 bits per sample.
    Like the reference, pixels is the raw byte image; a uint16 array is taken as little-endian 16-bit samples."""
    if not ((channels == 3 and bits_per_sample == 8) or (channels == 1 and bits_per_sample in (8, 16))):
        raise MicError(MIC_ERR_UNSUPPORTED, "compress_wsi: 8-bit RGB or 8/16-bit greyscale")
    arr = np.asarray(pixels)
    if arr.dtype == np.uint16:
        arr = arr.astype("<u2", copy=False)
    px = np.ascontiguousarray(arr).reshape(-1).view(np.uint8) if arr.dtype.itemsize == 2 else np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1)
    bpp = channels * (2 if bits_per_sample == 16 else 1)
    if px.size != width * height * bpp:
        raise MicError(MIC_ERR_ARGS, "compress_wsi")
    cap = px.size * 3 + (1 << 20)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().mic_hip_wsi_compress_ex(px.ctypes.data, width, height, channels, bits_per_sample, tile_w, tile_h, levels,
                                       out.ctypes.data, cap, C.byref(n))
    if rc:
        _raise(rc, "compress_wsi")
    return out[: n.value].tobytes()


def read_wsi_header(compressed):
    """ReadWSIHeader (wsicompress.go:299): dict with the format and the level table."""
    c = _bytes_arr(compressed)
    w, h, tw, th, nl = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    tot = C.c_uint64()
    rc = lib().mic_hip_wsi_info(c.ctypes.data, c.size, C.byref(w), C.byref(h), C.byref(tw), C.byref(th), C.byref(nl), C.byref(tot))
    if rc:
        _raise(rc, "read_wsi_header")
    ch, bps, ct = C.c_int(), C.c_int(), C.c_int()
    lib().mic_hip_wsi_format(c.ctypes.data, c.size, C.byref(ch), C.byref(bps), C.byref(ct))
    levels = []
    for i in range(nl.value):
        lw, lh, tx, ty = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib().mic_hip_wsi_level_info(c.ctypes.data, c.size, i, C.byref(lw), C.byref(lh), C.byref(tx), C.byref(ty))
        levels.append(dict(width=lw.value, height=lh.value, tiles_x=tx.value, tiles_y=ty.value))
    return dict(width=w.value, height=h.value, tile_width=tw.value, tile_height=th.value, total_tiles=tot.value,
                channels=ch.value, bits_per_sample=bps.value, color_transform=bool(ct.value), levels=levels)


def _wsi_shape(hdr, out: np.ndarray, w: int, h: int) -> np.ndarray:
    """bytes -> (h, w, 3) uint8 for RGB, (h, w) uint8 / uint16 for greyscale (uint16ToBytes, wsicompress.go:589-603)."""
    if hdr["channels"] == 3:
        return out[: w * h * 3].reshape(h, w, 3)
    if hdr["bits_per_sample"] == 16:
        return out[: w * h * 2].view("<u2").reshape(h, w)
    return out[: w * h].reshape(h, w)


def _wsi_bpp(hdr) -> int:
    return hdr["channels"] * (2 if hdr["bits_per_sample"] == 16 else 1)


def decompress_wsi_tile(compressed, level: int, tile_x: int, tile_y: int) -> np.ndarray:
    """DecompressWSITile (wsicompress.go:175): the tile cropped at the level's edge."""
    c = _bytes_arr(compressed)
    hdr = read_wsi_header(c)
    out = np.empty(hdr["tile_width"] * hdr["tile_height"] * _wsi_bpp(hdr), dtype=np.uint8)
    ow, oh = C.c_int(), C.c_int()
    rc = lib().mic_hip_wsi_decompress_tile(c.ctypes.data, c.size, level, tile_x, tile_y, out.ctypes.data, out.size, C.byref(ow), C.byref(oh))
    if rc:
        _raise(rc, "decompress_wsi_tile")
    return _wsi_shape(hdr, out, ow.value, oh.value).copy()


def decompress_wsi_region(compressed, level: int, x: int, y: int, w: int, h: int) -> np.ndarray:
    """DecompressWSIRegion (wsicompress.go:219): the rectangle clamped to the level."""
    c = _bytes_arr(compressed)
    hdr = read_wsi_header(c)
    out = np.empty(max(w, 0) * max(h, 0) * _wsi_bpp(hdr), dtype=np.uint8)
    ow, oh = C.c_int(), C.c_int()
    rc = lib().mic_hip_wsi_decompress_region(c.ctypes.data, c.size, level, x, y, w, h, out.ctypes.data, out.size, C.byref(ow), C.byref(oh))
    if rc:
        _raise(rc, "decompress_wsi_region")
    return _wsi_shape(hdr, out, ow.value, oh.value)


def decompress_wsi_level(compressed, level: int = 0) -> np.ndarray:
    c = _bytes_arr(compressed)
    hdr = read_wsi_header(c)
    lv = hdr["levels"][level]
    out = np.empty(lv["width"] * lv["height"] * _wsi_bpp(hdr), dtype=np.uint8)
    rc = lib().mic_hip_wsi_decompress_level(c.ctypes.data, c.size, level, out.ctypes.data, out.size)
    if rc:
        _raise(rc, "decompress_wsi_level")
    return _wsi_shape(hdr, out, lv["width"], lv["height"])


# ------------------------------------------------------------------ gradient predictor, PICA
def compress_single_frame_grad(pixels, width: int, height: int, max_value: int) -> bytes:
    """CompressSingleFrameGrad (multiframecompress.go:111)."""
    px = np.ascontiguousarray(pixels, dtype=np.uint16).reshape(-1)
    if px.size != width * height:
        raise MicError(MIC_ERR_ARGS, "compress_single_frame_grad")
    cap = _frame_bound(px.size)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().mic_hip_compress_frame_grad(px.ctypes.data, width, height, max_value, out.ctypes.data, cap, C.byref(n))
    if rc:
        _raise(rc, "compress_single_frame_grad")
    return out[: n.value].tobytes()


def decompress_single_frame_grad(compressed, width: int, height: int) -> np.ndarray:
    """DecompressSingleFrameGrad (multiframecompress.go:132)."""
    c = _bytes_arr(compressed)
    out = np.empty(width * height, dtype=np.uint16)
    rc = lib().mic_hip_decompress_frame_grad(c.ctypes.data, c.size, out.ctypes.data, width, height)
    if rc:
        _raise(rc, "decompress_single_frame_grad")
    return out.reshape(height, width)


def compress_parallel_strips_adaptive(pixels, width: int, height: int, max_value: int, num_strips: int) -> bytes:
    """CompressParallelStripsAdaptive (parallelstripsadaptive.go:54)."""
    px = np.ascontiguousarray(pixels, dtype=np.uint16).reshape(-1)
    if px.size != width * height:
        raise MicError(MIC_ERR_ARGS, "compress_parallel_strips_adaptive")
    cap = px.size * 4 + 135168 * max(1, num_strips) + 16 * max(1, num_strips) + 16
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().mic_hip_pica_compress(px.ctypes.data, width, height, max_value, num_strips, out.ctypes.data, cap, C.byref(n))
    if rc:
        _raise(rc, "compress_parallel_strips_adaptive")
    return out[: n.value].tobytes()


def decompress_parallel_strips_adaptive(compressed) -> np.ndarray:
    """DecompressParallelStripsAdaptive (parallelstripsadaptive.go:141): (height, width) uint16."""
    c = _bytes_arr(compressed)
    w, h, n = C.c_int(), C.c_int(), C.c_int()
    rc = lib().mic_hip_pica_info(c.ctypes.data, c.size, C.byref(w), C.byref(h), C.byref(n))
    if rc:
        _raise(rc, "decompress_parallel_strips_adaptive")
    out = np.empty(w.value * h.value, dtype=np.uint16)
    rc = lib().mic_hip_pica_decompress(c.ctypes.data, c.size, out.ctypes.data, w.value, h.value)
    if rc:
        _raise(rc, "decompress_parallel_strips_adaptive")
    return out.reshape(h.value, w.value)


# ------------------------------------------------------------------ single-frame RGB, MIC1 / MICR files
def compress_rgb(rgb, width: int, height: int, container: bool = False) -> bytes:
    """CompressRGB (rgbcompress.go:25); container=True wraps it as a MICR file (cmd/mic-compress/main.go:62-91)."""
    px = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1)
    if px.size != width * height * 3:
        raise MicError(MIC_ERR_ARGS, "compress_rgb")
    cap = px.size * 4 + 4096
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    fn = lib().mic_hip_micr_compress if container else lib().mic_hip_rgb_compress
    rc = fn(px.ctypes.data, width, height, out.ctypes.data, cap, C.byref(n))
    if rc:
        _raise(rc, "compress_rgb")
    return out[: n.value].tobytes()


def decompress_rgb(compressed, width: int = 0, height: int = 0) -> np.ndarray:
    """DecompressRGB (rgbcompress.go:31) when width / height are given, else a MICR file: (h, w, 3) uint8."""
    c = _bytes_arr(compressed)
    if width and height:
        out = np.empty(width * height * 3, dtype=np.uint8)
        rc = lib().mic_hip_rgb_decompress(c.ctypes.data, c.size, width, height, out.ctypes.data, out.size)
    else:
        w, h = C.c_int(), C.c_int()
        rc = lib().mic_hip_micr_info(c.ctypes.data, c.size, C.byref(w), C.byref(h))
        if rc:
            _raise(rc, "decompress_rgb")
        width, height = w.value, h.value
        out = np.empty(width * height * 3, dtype=np.uint8)
        rc = lib().mic_hip_micr_decompress(c.ctypes.data, c.size, out.ctypes.data, out.size)
    if rc:
        _raise(rc, "decompress_rgb")
    return out.reshape(height, width, 3)


def write_mic1(pixels, width: int, height: int, max_value: int, nstates: int = 2) -> bytes:
    """The CLI's single-frame .mic file (writeMicFile, cmd/mic-compress/main.go:26-59)."""
    px = np.ascontiguousarray(pixels, dtype=np.uint16).reshape(-1)
    if px.size != width * height:
        raise MicError(MIC_ERR_ARGS, "write_mic1")
    cap = _frame_bound(px.size) + 20
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().mic_hip_mic1_compress(px.ctypes.data, width, height, max_value, nstates, out.ctypes.data, cap, C.byref(n))
    if rc:
        _raise(rc, "write_mic1")
    return out[: n.value].tobytes()


def read_mic1(compressed) -> np.ndarray:
    c = _bytes_arr(compressed)
    w, h = C.c_int(), C.c_int()
    rc = lib().mic_hip_mic1_info(c.ctypes.data, c.size, C.byref(w), C.byref(h))
    if rc:
        _raise(rc, "read_mic1")
    out = np.empty(w.value * h.value, dtype=np.uint16)
    rc = lib().mic_hip_mic1_decompress(c.ctypes.data, c.size, out.ctypes.data, out.size)
    if rc:
        _raise(rc, "read_mic1")
    return out.reshape(h.value, w.value)


# ------------------------------------------------------------------ device-resident sessions
class Session:
    """mic_hip_session: encode/decode units that already live in HBM.  Device pointers are
    plain integers (e.g. ``torch.Tensor.data_ptr()``); torch itself is not needed here."""

    def __init__(self, max_units: int, max_px_per_unit: int, device: Optional[int] = None):
        """device = None: the default session's device (mic_hip_set_device); an int: that HIP device -- one host process can
        hold a session per GPU (mic_hip_session_create_on)."""
        self._h = C.c_void_p()
        if device is None:
            rc = lib().mic_hip_session_create(C.byref(self._h), max_units, max_px_per_unit)
        else:
            rc = lib().mic_hip_session_create_on(int(device), C.byref(self._h), max_units, max_px_per_unit)
        if rc:
            _raise(rc, "session_create")
        self._n = 0

    @property
    def device(self) -> int:
        return lib().mic_hip_session_device(self._h)

    def close(self):
        if self._h:
            lib().mic_hip_session_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return lib().mic_hip_session_stream(self._h) or 0

    def set_timing(self, on):
        """False / 0: off; True / 1: per-kernel HIP-event timing of the next launch chain; 2: summed over every launch chain
        until the next set_timing (calls that run several chains: slabs of a slide, a wavelet pass)."""
        lib().mic_hip_session_set_timing(self._h, int(on))

    def last_timings(self):
        names = (C.c_char_p * 96)()
        ms = (C.c_float * 96)()
        k = lib().mic_hip_session_last_timings(self._h, names, ms, 96)
        return [(names[i].decode(), float(ms[i])) for i in range(k)]

    @staticmethod
    def make_units(units: Sequence[Tuple[int, int, int, int, int]]):
        arr = (Unit * len(units))()
        for i, (off, w, h, mv, ns) in enumerate(units):
            arr[i].px_offset = off; arr[i].width = w; arr[i].height = h; arr[i].max_value = mv; arr[i].nstates = ns
        return arr

    def encode_enqueue(self, d_pixels: int, units):
        rc = lib().mic_hip_session_encode_enqueue(self._h, d_pixels, units, len(units))
        if rc:
            _raise(rc, "session_encode_enqueue")
        self._n = len(units)

    # (results land in numpy arrays the library writes straight into: building them element by element from ctypes arrays -- and the
    # offset table of decode_enqueue from a Python list -- was 0.4 ms of host time per step of 2304 units with the device idle)
    def encode_finish(self):
        n = self._n
        offs = np.empty(n + 1, dtype=np.uint64); st = np.empty(n, dtype=np.int32); ns = np.empty(n, dtype=np.int32)
        d = C.c_void_p()
        rc = lib().mic_hip_session_encode_finish(self._h, C.byref(d), offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                 st.ctypes.data_as(C.POINTER(C.c_int32)), ns.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc:
            _raise(rc, "session_encode_finish")
        return d.value, offs, st, ns

    def decode_enqueue(self, d_blobs: int, offsets: np.ndarray, units, d_pixels_out: int):
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        if offs.size < len(units) + 1:
            raise ValueError("decode_enqueue: offsets must hold len(units) + 1 entries")
        rc = lib().mic_hip_session_decode_enqueue(self._h, d_blobs, offs.ctypes.data_as(C.POINTER(C.c_uint64)), units, len(units), d_pixels_out)
        if rc:
            _raise(rc, "session_decode_enqueue")
        self._n = len(units)

    def decode_finish(self) -> np.ndarray:
        st = np.empty(self._n, dtype=np.int32)
        rc = lib().mic_hip_session_decode_finish(self._h, st.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc:
            _raise(rc, "session_decode_finish")
        return st

    # ---- WaveletV2 on device-resident frames (waveletfsecompressu16.go:303-534) -------------------------------------------
    def wavelet_v2_encode(self, d_frames: int, nframes: int, rows: int, cols: int, levels: int = 5):
        """-> (device pointer of the packed header-less streams, offsets[nframes + 1], status[nframes], levels applied)"""
        d = C.c_void_p(); offs = (C.c_uint64 * (nframes + 1))(); st = (C.c_int32 * nframes)(); ap = C.c_int(0)
        rc = lib().mic_hip_session_wavelet_v2_encode(self._h, d_frames, nframes, rows, cols, levels, C.byref(d), offs, st, C.byref(ap))
        if rc:
            _raise(rc, "session_wavelet_v2_encode")
        return d.value, np.array(offs[:], dtype=np.uint64), np.array(st[:], dtype=np.int32), ap.value

    def wavelet_v2_decode(self, d_streams: int, offsets: np.ndarray, nframes: int, rows: int, cols: int, levels: int, d_pixels_out: int) -> np.ndarray:
        offs = (C.c_uint64 * len(offsets))(*[int(v) for v in offsets]); st = (C.c_int32 * nframes)()
        rc = lib().mic_hip_session_wavelet_v2_decode(self._h, d_streams, offs, nframes, rows, cols, levels, d_pixels_out, st)
        if rc:
            _raise(rc, "session_wavelet_v2_decode")
        return np.array(st[:], dtype=np.int32)

    # ---- MIC3 on a device-resident slide (wsicompress.go:27-171) ---------------------------------------------------------------
    def wsi_encode(self, d_pixels: int, width: int, height: int, channels: int = 3, bits_per_sample: int = 8,
                   tile_w: int = 0, tile_h: int = 0, levels: int = 0) -> Tuple[int, int]:
        """-> (tiles, size of the MIC3 file the coded planes make); the planes stay in the session"""
        tt = C.c_uint64(0); cb = C.c_uint64(0)
        rc = lib().mic_hip_session_wsi_encode(self._h, d_pixels, width, height, channels, bits_per_sample, tile_w, tile_h, levels, C.byref(tt), C.byref(cb))
        if rc:
            _raise(rc, "session_wsi_encode")
        self._wsi_bytes = cb.value
        return tt.value, cb.value

    def wsi_write(self, out: Optional[np.ndarray] = None):
        """WriteMIC3 around the session's coded planes: the file CompressWSI returns, as bytes -- or, with `out` (a uint8 array of
        the caller's, ordinary or pinned), written there: returns its length"""
        buf = out if out is not None else np.empty(self._wsi_bytes + 64, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = lib().mic_hip_session_wsi_write(self._h, buf.ctypes.data, buf.size, C.byref(n))
        if rc:
            _raise(rc, "session_wsi_write")
        return n.value if out is not None else buf[: n.value].tobytes()

    def workspace_bytes(self) -> Tuple[int, bool]:
        """(device bytes the session holds, whether a batch has needed the tier-2 slabs)"""
        t = C.c_int(0)
        n = lib().mic_hip_session_workspace_bytes(self._h, C.byref(t))
        return int(n), bool(t.value)

    def wsi_payload(self, total_tiles: int) -> Tuple[int, int, np.ndarray]:
        """-> (device address of the container's payload, its size, tile lengths in container order); valid until the next wsi call"""
        d = C.c_void_p(0); nb = C.c_uint64(0); lens = np.zeros(max(total_tiles, 1), dtype=np.uint64)
        rc = lib().mic_hip_session_wsi_payload(self._h, C.byref(d), C.byref(nb), lens.ctypes.data, lens.size)
        if rc:
            _raise(rc, "session_wsi_payload")
        return d.value, nb.value, lens[:total_tiles].astype(np.int64)

    def wsi_levels(self) -> List[Tuple[int, int]]:
        n = C.c_int(0); w = (C.c_int * 32)(); h = (C.c_int * 32)()
        rc = lib().mic_hip_session_wsi_levels(self._h, C.byref(n), w, h, 32)
        if rc:
            _raise(rc, "session_wsi_levels")
        return [(w[i], h[i]) for i in range(n.value)]

    def wsi_decode_level(self, level: int, d_pixels_out: int, out_cap: int):
        rc = lib().mic_hip_session_wsi_decode_level(self._h, level, d_pixels_out, out_cap)
        if rc:
            _raise(rc, "session_wsi_decode_level")

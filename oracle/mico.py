"""ctypes binding of the CPU oracle (oracle/libmic_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmic_oracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "libmic_ref.so")

OK, ERR_ARGS, ERR_USE_RLE, ERR_CAPACITY, ERR_CORRUPT, ERR_INCOMPRESSIBLE = 0, -1, -3, -5, -6, -10

_lib = None


def build(ref: bool = False) -> None:
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["ref"] if ref else []))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.mico_fnv1a64.restype = C.c_uint64
        _lib.mico_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data)


def fnv1a64(b) -> int:
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    return int(lib().mico_fnv1a64(_p(a), a.size))


def delta_rle_compress(px: np.ndarray, max_value: int) -> np.ndarray:
    px = np.ascontiguousarray(px, dtype=np.uint16)
    h, w = px.shape
    out = np.empty(4 * px.size + 16, dtype=np.uint16)
    n = C.c_size_t()
    rc = lib().mico_delta_rle_compress(_p(px), w, h, C.c_uint16(max_value), _p(out), C.c_size_t(out.size), C.byref(n))
    if rc:
        raise RuntimeError(f"mico_delta_rle_compress rc={rc}")
    return out[: n.value].copy()


def delta_symbols(px: np.ndarray, max_value: int) -> np.ndarray:
    px = np.ascontiguousarray(px, dtype=np.uint16)
    h, w = px.shape
    out = np.empty(2 * px.size + 16, dtype=np.uint16)
    n = C.c_size_t()
    rc = lib().mico_delta_symbols(_p(px), w, h, C.c_uint16(max_value), _p(out), C.c_size_t(out.size), C.byref(n))
    if rc:
        raise RuntimeError(f"mico_delta_symbols rc={rc}")
    return out[: n.value].copy()


def fse_compress(sym: np.ndarray, nstates: int):
    """returns (rc, bytes)"""
    sym = np.ascontiguousarray(sym, dtype=np.uint16)
    out = np.empty(sym.size * 2 + 200000, dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().mico_fse_compress(_p(sym), C.c_size_t(sym.size), nstates, _p(out), C.c_size_t(out.size), C.byref(n))
    return rc, (out[: n.value].tobytes() if rc == 0 else b"")


def fse_decompress_auto(b: bytes, cap: int):
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    out = np.empty(cap, dtype=np.uint16)
    n = C.c_size_t()
    rc = lib().mico_fse_decompress_auto(_p(a), C.c_size_t(a.size), _p(out), C.c_size_t(cap), C.byref(n))
    return rc, (out[: n.value].copy() if rc == 0 else None)


def compress_single_frame(px: np.ndarray, max_value: int, nstates: int = 2):
    """returns (rc, bytes)"""
    px = np.ascontiguousarray(px, dtype=np.uint16)
    h, w = px.shape
    out = np.empty(px.size * 4 + 200000, dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().mico_compress_single_frame(_p(px), w, h, C.c_uint16(max_value), nstates, _p(out), C.c_size_t(out.size), C.byref(n))
    return rc, (out[: n.value].tobytes() if rc == 0 else b"")


def decompress_single_frame(b: bytes, w: int, h: int):
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    out = np.empty((h, w), dtype=np.uint16)
    rc = lib().mico_decompress_single_frame(_p(a), C.c_size_t(a.size), _p(out), w, h)
    return rc, (out if rc == 0 else None)


def pics_compress(px: np.ndarray, max_value: int, num_strips: int, nstates: int = 2):
    px = np.ascontiguousarray(px, dtype=np.uint16)
    h, w = px.shape
    out = np.empty(px.size * 4 + 200000 * max(1, num_strips), dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().mico_pics_compress(_p(px), w, h, C.c_uint16(max_value), num_strips, nstates, _p(out), C.c_size_t(out.size), C.byref(n))
    return rc, (out[: n.value].tobytes() if rc == 0 else b"")


def pics_decompress(b: bytes):
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    w, h = C.c_int(), C.c_int()
    rc = lib().mico_pics_decompress(_p(a), C.c_size_t(a.size), None, C.c_size_t(0), C.byref(w), C.byref(h))
    if rc:
        return rc, None
    out = np.empty((h.value, w.value), dtype=np.uint16)
    rc = lib().mico_pics_decompress(_p(a), C.c_size_t(a.size), _p(out), C.c_size_t(out.size), C.byref(w), C.byref(h))
    return rc, (out if rc == 0 else None)


def mic2_compress(frames: np.ndarray, max_value: int, temporal: bool = False):
    fr = np.ascontiguousarray(frames, dtype=np.uint16)
    n_, h, w = fr.shape
    out = np.empty(fr.size * 4 + 200000 * n_, dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().mico_mic2_compress(_p(fr), w, h, n_, C.c_uint16(max_value), 1 if temporal else 0, _p(out), C.c_size_t(out.size), C.byref(n))
    return rc, (out[: n.value].tobytes() if rc == 0 else b"")


def mic2_decompress(b: bytes):
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    w, h, n_ = C.c_int(), C.c_int(), C.c_int()
    rc = lib().mico_mic2_decompress(_p(a), C.c_size_t(a.size), None, C.c_size_t(0), C.byref(w), C.byref(h), C.byref(n_))
    if rc:
        return rc, None
    out = np.empty((n_.value, h.value, w.value), dtype=np.uint16)
    rc = lib().mico_mic2_decompress(_p(a), C.c_size_t(a.size), _p(out), C.c_size_t(out.size), C.byref(w), C.byref(h), C.byref(n_))
    return rc, (out if rc == 0 else None)

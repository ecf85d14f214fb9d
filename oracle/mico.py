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


def fse_compress_tl(sym: np.ndarray, nstates: int, table_log: int):
    """FSECompressU16* with ScratchU16.TableLog = table_log (fseu16.go:101-102)"""
    sym = np.ascontiguousarray(sym, dtype=np.uint16)
    out = np.empty(sym.size * 2 + 200000, dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().mico_fse_compress_tl(_p(sym), C.c_size_t(sym.size), nstates, table_log, _p(out), C.c_size_t(out.size), C.byref(n))
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


def grad_delta_rle_compress(px: np.ndarray, max_value: int) -> np.ndarray:
    px = np.ascontiguousarray(px, dtype=np.uint16)
    h, w = px.shape
    out = np.empty(4 * px.size + 16, dtype=np.uint16)
    n = C.c_size_t()
    rc = lib().mico_grad_delta_rle_compress(_p(px), w, h, C.c_uint16(max_value), _p(out), C.c_size_t(out.size), C.byref(n))
    if rc:
        raise RuntimeError(f"mico_grad_delta_rle_compress rc={rc}")
    return out[: n.value].copy()


def grad_delta_rle_decompress(tok: np.ndarray, w: int, h: int):
    tok = np.ascontiguousarray(tok, dtype=np.uint16)
    out = np.empty((h, w), dtype=np.uint16)
    rc = lib().mico_grad_delta_rle_decompress(_p(tok), C.c_size_t(tok.size), w, h, _p(out))
    return rc, (out if rc == 0 else None)


def compress_single_frame_grad(px: np.ndarray, max_value: int):
    px = np.ascontiguousarray(px, dtype=np.uint16)
    h, w = px.shape
    out = np.empty(px.size * 4 + 200000, dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().mico_compress_single_frame_grad(_p(px), w, h, C.c_uint16(max_value), _p(out), C.c_size_t(out.size), C.byref(n))
    return rc, (out[: n.value].tobytes() if rc == 0 else b"")


def decompress_single_frame_grad(b: bytes, w: int, h: int):
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    out = np.empty((h, w), dtype=np.uint16)
    rc = lib().mico_decompress_single_frame_grad(_p(a), C.c_size_t(a.size), _p(out), w, h)
    return rc, (out if rc == 0 else None)


def pica_boundaries(px: np.ndarray, num_strips: int):
    px = np.ascontiguousarray(px, dtype=np.uint16)
    h, w = px.shape
    starts = np.zeros(max(num_strips, 1), dtype=np.int32)
    n = lib().mico_pica_boundaries(_p(px), w, h, num_strips, _p(starts))
    return starts[:n].tolist()


def pica_compress(px: np.ndarray, max_value: int, num_strips: int):
    px = np.ascontiguousarray(px, dtype=np.uint16)
    h, w = px.shape
    out = np.empty(px.size * 4 + 200000 * max(1, num_strips), dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().mico_pica_compress(_p(px), w, h, C.c_uint16(max_value), num_strips, _p(out), C.c_size_t(out.size), C.byref(n))
    return rc, (out[: n.value].tobytes() if rc == 0 else b"")


def pica_decompress(b: bytes):
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    w, h = C.c_int(), C.c_int()
    rc = lib().mico_pica_decompress(_p(a), C.c_size_t(a.size), None, C.c_size_t(0), C.byref(w), C.byref(h))
    if rc:
        return rc, None
    out = np.empty((h.value, w.value), dtype=np.uint16)
    rc = lib().mico_pica_decompress(_p(a), C.c_size_t(a.size), _p(out), C.c_size_t(out.size), C.byref(w), C.byref(h))
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


def rle_compress(sym: np.ndarray, max_value: int) -> np.ndarray:
    sym = np.ascontiguousarray(sym, dtype=np.uint16)
    out = np.empty(2 * sym.size + 64, dtype=np.uint16)
    n = C.c_size_t()
    rc = lib().mico_rle_compress(_p(sym), C.c_size_t(sym.size), C.c_uint16(max_value), _p(out), C.c_size_t(out.size), C.byref(n))
    if rc:
        raise RuntimeError(f"mico_rle_compress rc={rc}")
    return out[: n.value].copy()


def wt53_forward(data: np.ndarray, levels: int):
    """in-place multi-level forward transform of an int32 (rows, cols) array; returns levels applied"""
    assert data.dtype == np.int32 and data.flags.c_contiguous
    rows, cols = data.shape
    applied = C.c_int()
    rc = lib().mico_wt53_forward(_p(data), rows, cols, levels, C.byref(applied))
    if rc:
        raise RuntimeError(f"mico_wt53_forward rc={rc}")
    return applied.value


def wt53_inverse(data: np.ndarray, levels: int) -> None:
    assert data.dtype == np.int32 and data.flags.c_contiguous
    rows, cols = data.shape
    rc = lib().mico_wt53_inverse(_p(data), rows, cols, levels)
    if rc:
        raise RuntimeError(f"mico_wt53_inverse rc={rc}")


def wavelet_v2_compress(px: np.ndarray, max_value: int, levels: int = 5):
    px = np.ascontiguousarray(px, dtype=np.uint16)
    rows, cols = px.shape
    out = np.empty(px.size * 6 + 200000, dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().mico_wavelet_v2_compress(_p(px), rows, cols, C.c_uint16(max_value), levels, _p(out), C.c_size_t(out.size), C.byref(n))
    return rc, (out[: n.value].tobytes() if rc == 0 else b"")


def wavelet_v2_decompress(b: bytes):
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    r, c = C.c_int(), C.c_int()
    rc = lib().mico_wavelet_v2_decompress(_p(a), C.c_size_t(a.size), None, C.c_size_t(0), C.byref(r), C.byref(c))
    if rc:
        return rc, None
    out = np.empty((r.value, c.value), dtype=np.uint16)
    rc = lib().mico_wavelet_v2_decompress(_p(a), C.c_size_t(a.size), _p(out), C.c_size_t(out.size), C.byref(r), C.byref(c))
    return rc, (out if rc == 0 else None)


def ycocgr_forward(rgb: np.ndarray):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, 3)
    n = rgb.shape[0]
    y = np.empty(n, np.uint16); co = np.empty(n, np.uint16); cg = np.empty(n, np.uint16)
    lib().mico_ycocgr_forward(_p(rgb), n, _p(y), _p(co), _p(cg))
    return y, co, cg


def ycocgr_inverse(y, co, cg) -> np.ndarray:
    n = y.size
    rgb = np.empty((n, 3), np.uint8)
    lib().mico_ycocgr_inverse(_p(np.ascontiguousarray(y)), _p(np.ascontiguousarray(co)), _p(np.ascontiguousarray(cg)), n, _p(rgb))
    return rgb


def wsi_compress_tile(rgb: np.ndarray):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    th, tw, _ = rgb.shape
    out = np.empty(rgb.size * 4 + 4096, dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().mico_wsi_compress_tile(_p(rgb), tw, th, _p(out), C.c_size_t(out.size), C.byref(n))
    return rc, (out[: n.value].tobytes() if rc == 0 else b"")


def wsi_decompress_tile(b: bytes, tw: int, th: int):
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    out = np.empty((th, tw, 3), dtype=np.uint8)
    rc = lib().mico_wsi_decompress_tile(_p(a), C.c_size_t(a.size), tw, th, _p(out))
    return rc, (out if rc == 0 else None)


def wsi_compress(rgb: np.ndarray, tile_w: int = 256, tile_h: int = 256, levels: int = 0):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    out = np.empty(rgb.size * 6 + (1 << 20), dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().mico_wsi_compress(_p(rgb), w, h, tile_w, tile_h, levels, _p(out), C.c_size_t(out.size), C.byref(n))
    return rc, (out[: n.value].tobytes() if rc == 0 else b"")


def micr_write(rgb: np.ndarray):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    out = np.empty(rgb.size * 4 + 4096, dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().mico_micr_write(_p(rgb), w, h, _p(out), C.c_size_t(out.size), C.byref(n))
    return rc, (out[: n.value].tobytes() if rc == 0 else b"")


def micr_read(b: bytes):
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    w, h = int.from_bytes(bytes(b[4:8]), "little"), int.from_bytes(bytes(b[8:12]), "little")
    out = np.empty((h, w, 3), dtype=np.uint8)
    cw, ch = C.c_int(), C.c_int()
    rc = lib().mico_micr_read(_p(a), C.c_size_t(a.size), _p(out), C.c_size_t(out.size), C.byref(cw), C.byref(ch))
    return rc, (out if rc == 0 else None)


def mic1_write(px: np.ndarray, max_value: int, nstates: int = 2):
    px = np.ascontiguousarray(px, dtype=np.uint16)
    h, w = px.shape
    out = np.empty(px.size * 2 + 8192, dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().mico_mic1_write(_p(px), w, h, C.c_uint16(max_value), nstates, _p(out), C.c_size_t(out.size), C.byref(n))
    return rc, (out[: n.value].tobytes() if rc == 0 else b"")


def mic1_read(b: bytes):
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    w, h = int.from_bytes(bytes(b[4:8]), "little"), int.from_bytes(bytes(b[8:12]), "little")
    out = np.empty((h, w), dtype=np.uint16)
    cw, ch = C.c_int(), C.c_int()
    rc = lib().mico_mic1_read(_p(a), C.c_size_t(a.size), _p(out), C.c_size_t(out.size), C.byref(cw), C.byref(ch))
    return rc, (out if rc == 0 else None)


def wsi_compress_grey(img: np.ndarray, tile_w: int = 256, tile_h: int = 256, levels: int = 0):
    """CompressWSI with channels=1; uint8 -> 8 bits per sample, uint16 -> 16 (little-endian bytes)."""
    bps = 16 if img.dtype == np.uint16 else 8
    img = np.ascontiguousarray(img, dtype="<u2" if bps == 16 else np.uint8)
    h, w = img.shape
    out = np.empty(img.nbytes * 4 + (1 << 20), dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().mico_wsi_compress_ex(_p(img), w, h, 1, bps, tile_w, tile_h, levels, _p(out), C.c_size_t(out.size), C.byref(n))
    return rc, (out[: n.value].tobytes() if rc == 0 else b"")


def wsi_decompress_tile_at(b: bytes, level: int, tx: int, ty: int, tile_w: int = 256, tile_h: int = 256):
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    channels, bps = int(a[24]) | (int(a[25]) << 8), int(a[26])
    if channels == 1:
        out = np.empty(tile_w * tile_h * (2 if bps == 16 else 1), dtype=np.uint8)
        tw, th = C.c_int(), C.c_int()
        rc = lib().mico_wsi_decompress_tile_at(_p(a), C.c_size_t(a.size), level, tx, ty, _p(out), C.c_size_t(out.size), C.byref(tw), C.byref(th))
        if rc:
            return rc, None
        n = tw.value * th.value
        px = out[: n * 2].view("<u2") if bps == 16 else out[:n]
        return rc, px.reshape(th.value, tw.value).copy()
    out = np.empty(tile_w * tile_h * 3, dtype=np.uint8)
    tw, th = C.c_int(), C.c_int()
    rc = lib().mico_wsi_decompress_tile_at(_p(a), C.c_size_t(a.size), level, tx, ty, _p(out), C.c_size_t(out.size), C.byref(tw), C.byref(th))
    if rc:
        return rc, None
    return rc, out[: tw.value * th.value * 3].reshape(th.value, tw.value, 3).copy()

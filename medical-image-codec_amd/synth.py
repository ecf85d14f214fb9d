"""Synthetic stand-ins for the reference's absent test images (XR, CR, MG
blobs are missing; Go math/rand generators are not reproducible outside Go).

Everything here is closed-form + a counter-based hash PRNG (splitmix64 of the pixel index),
so the same pixels come out on any machine and any numpy.  Generators are committed, data
is not.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_u64(n: int, seed: int) -> np.ndarray:
    """n pseudo-random u64 values: splitmix64(seed*2^32 + i)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))
        return _splitmix64(idx)


def approx_gauss(n: int, seed: int) -> np.ndarray:
    """Zero-mean, unit-variance, bell-shaped noise: sum of 4 uniform 16-bit fields."""
    h = hash_u64(n, seed)
    s = np.zeros(n, dtype=np.float64)
    for k in range(4):
        s += ((h >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.float64)
    s = (s - 4 * 32767.5) / (np.sqrt(4.0 / 12.0) * 65536.0)
    return s


def _xr_field(cols: int, rows: int, depth: int):
    """The noise-free part of xr_like: (f * maxv, sqrt(f + 0.1)) as float64 (rows, cols)."""
    maxv = (1 << depth) - 1
    y = np.linspace(-1.0, 1.0, rows, dtype=np.float64)[:, None]
    x = np.linspace(-1.0, 1.0, cols, dtype=np.float64)[None, :]
    f = 0.55 + 0.25 * np.cos(2.1 * x + 0.3) * np.cos(1.7 * y - 0.2)
    f += 0.18 * np.exp(-((x - 0.2) ** 2 / 0.08 + (y + 0.1) ** 2 / 0.3))
    f -= 0.22 * np.exp(-((x + 0.35) ** 2 / 0.02 + (y - 0.2) ** 2 / 0.5))
    f += 0.03 * np.sin(23.0 * x) * np.sin(17.0 * y)
    f = np.clip(f, 0.02, 0.98)
    return f * maxv, np.sqrt(f + 0.1)


def _xr_marks(img, cols: int, rows: int, maxv: int):
    """collimator border (constant -> RLE runs) and a few saturated markers (-> escape path); img: (..., rows, cols)"""
    b = max(4, rows // 64)
    img[..., :b, :] = 0
    img[..., -b:, :] = 0
    img[..., :, : max(4, cols // 80)] = 0
    img[..., :, -max(4, cols // 80):] = 0
    for k in range(6):                                                   # lead markers: saturated blobs, sharp edges -> |diff| >= T escapes
        cy = int(rows * (0.15 + 0.12 * k)); cx = int(cols * (0.1 + 0.14 * k))
        img[..., cy:cy + 24, cx:cx + 24] = maxv
    return img


XR_NOISE_PUBLISHED_RATIO = 167.0   # xr_like noise at which PICS-8 of the 2577 x 2048 frame codes at ~1.755, the reference's published XR ratio


def xr_like(cols: int = 2577, rows: int = 2048, depth: int = 12, seed: int = 1,
            noise: float = 18.0) -> np.ndarray:
    """'XR-like' frame of the reference's XR shape (cols 2577 x rows 2048,
    fseu16_test.go:32): smooth anatomy-like field + signal-dependent noise, collimator
    border (constant -> RLE runs) and a few saturated markers (-> escape path).
    noise = 18 codes at ratio ~2.65 (PICS-8); XR_NOISE_PUBLISHED_RATIO at the reference's published 1.755."""
    maxv = (1 << depth) - 1
    fm, sq = _xr_field(cols, rows, depth)
    g = approx_gauss(rows * cols, seed).reshape(rows, cols)
    img = fm + g * noise * (maxv / 4095.0) * sq
    img = np.clip(np.rint(img), 0, maxv).astype(np.uint16)
    return _xr_marks(img, cols, rows, maxv)


def xr_like_batch_torch(n: int, cols: int = 2577, rows: int = 2048, depth: int = 12, seed0: int = 1, noise: float = 18.0,
                        device="cuda", chunk: int = 8):
    """n DISTINCT frames, frame i == xr_like(cols, rows, depth, seed0 + i, noise) bit for bit, made on the device (torch is
    plumbing here: the same splitmix64 hash in int64 arithmetic, the same float64 operations in the same order -- IEEE add,
    multiply, divide and round-half-even give the numpy result).  Returns an int16 tensor (n, rows, cols) holding the u16 bits."""
    import torch
    maxv = (1 << depth) - 1
    fm_h, sq_h = _xr_field(cols, rows, depth)
    fm = torch.from_numpy(fm_h).to(device).reshape(-1)
    sq = torch.from_numpy(sq_h).to(device).reshape(-1)
    npx = rows * cols
    idx = torch.arange(npx, dtype=torch.int64, device=device)
    out = torch.empty((n, rows, cols), dtype=torch.int16, device=device)

    def s64(v):                                                          # python int -> the int64 with the same 64 bits
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >> 63 else v

    def lsr(z, k):                                                       # logical shift right of int64 bit patterns
        return (z >> k) & ((1 << (64 - k)) - 1)

    scale = float(np.sqrt(4.0 / 12.0) * 65536.0)
    amp = maxv / 4095.0
    for i0 in range(0, n, chunk):
        k = min(chunk, n - i0)
        seeds = torch.arange(seed0 + i0, seed0 + i0 + k, dtype=torch.int64, device=device)[:, None]
        z = idx[None, :] + (seeds << 32) + s64(0x9E3779B97F4A7C15)
        z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
        h = z ^ lsr(z, 31)
        s = torch.zeros((k, npx), dtype=torch.float64, device=device)
        for j in range(4):
            s += ((h >> (16 * j)) & 0xFFFF).to(torch.float64)
        g = (s - 4 * 32767.5) / scale
        img = fm[None, :] + g * noise * amp * sq[None, :]
        img = torch.clamp(torch.round(img), 0, maxv).to(torch.int32).reshape(k, rows, cols)
        img = _xr_marks(img, cols, rows, maxv)
        out[i0:i0 + k] = img.to(torch.int16)                             # (wraps above 32767: the u16 bit pattern)
        del z, h, s, g, img
    return out


def cr_like(cols: int = 1760, rows: int = 2140, depth: int = 12, seed: int = 2) -> np.ndarray:
    """'CR-like' frame of the reference's CR shape (cols 1760 x rows 2140,
    fseu16_test.go:31): smoother, lower noise than xr_like (ratio ~3.7 in the reference)."""
    return xr_like(cols=cols, rows=rows, depth=depth, seed=seed, noise=5.0)


def ct_stack(frames: int = 512, size: int = 512, depth: int = 12, seed: int = 3) -> np.ndarray:
    """CT-like stack (frames x size x size), slowly varying along z: air background
    (constant), elliptical body with organs, quantum noise."""
    maxv = (1 << depth) - 1
    y = np.linspace(-1.0, 1.0, size, dtype=np.float64)[:, None]
    x = np.linspace(-1.0, 1.0, size, dtype=np.float64)[None, :]
    out = np.empty((frames, size, size), dtype=np.uint16)
    for z in range(frames):
        t = z / max(frames - 1, 1)
        a, b = 0.75 + 0.1 * np.sin(3.0 * t), 0.55 + 0.08 * np.cos(2.0 * t)
        body = ((x / a) ** 2 + (y / b) ** 2) < 1.0
        f = np.where(body, 0.26, 0.0)
        f = f + np.where(((x - 0.25) ** 2 + (y + 0.05 + 0.1 * t) ** 2) < 0.04, 0.02, 0.0)
        f = f + np.where(((x + 0.3) ** 2 / 0.03 + (y - 0.1) ** 2 / 0.06) < 1.0, -0.2 * body, 0.0)
        f = f + np.where(((x) ** 2 + (y - 0.42) ** 2) < 0.006, 0.35, 0.0)
        g = approx_gauss(size * size, seed * 100003 + z).reshape(size, size)
        img = f * maxv + g * 9.0 * body
        out[z] = np.clip(np.rint(img), 0, maxv).astype(np.uint16)
    return out


def wsi_like(width: int, height: int, seed: int = 4) -> np.ndarray:
    """H&E-like RGB slide (height x width x 3, uint8) in the spirit of wsi_test.go:71-122:
    white background + textured tissue discs."""
    yy = np.arange(height, dtype=np.float64)[:, None]
    xx = np.arange(width, dtype=np.float64)[None, :]
    img = np.full((height, width, 3), 255, dtype=np.uint8)
    h = hash_u64(height * width, seed).reshape(height, width)
    n0 = ((h & np.uint64(0xFF)).astype(np.int32) - 128)
    n1 = (((h >> np.uint64(8)) & np.uint64(0xFF)).astype(np.int32) - 128)
    for k, (cx, cy, r) in enumerate(((0.35, 0.4, 0.28), (0.68, 0.62, 0.2))):
        mask = ((xx - cx * width) ** 2 + (yy - cy * height) ** 2) < (r * min(width, height)) ** 2
        tex = 20.0 * np.sin(xx / 7.0 + k) * np.cos(yy / 9.0)
        rch = np.clip(200 + tex + n0 / 10.0, 0, 255)
        gch = np.clip(120 + tex * 0.6 + n1 / 12.0, 0, 255)
        bch = np.clip(170 + tex * 0.8 + n0 / 16.0, 0, 255)
        for c, ch in enumerate((rch, gch, bch)):
            plane = img[:, :, c]
            plane[mask] = ch.astype(np.uint8)[mask]
    return img


def closed_form(n: int, mul: int = 131, add: int = 7, mod: int = 65536) -> np.ndarray:
    """The reference wavelet tests' closed-form inputs: (i*131+7)%65536, (i*97+13)%4096
    (waveletu16_test.go:190-248)."""
    i = np.arange(n, dtype=np.int64)
    return ((i * mul + add) % mod).astype(np.uint16)


def wsi_slide_band(width: int, height: int, y0: int, y1: int, seed: int = 4) -> np.ndarray:
    """Rows [y0, y1) of wsi_slide(width, height, seed): 32-bit integer arithmetic only (a band of a 32768 x 32768 slide
    must come out in a fraction of a second), pixel (x, y) a pure function of (x, y, width, height, seed)."""
    yy = np.arange(y0, y1, dtype=np.int64)[:, None]
    xx = np.arange(width, dtype=np.int64)[None, :]
    m = min(width, height)
    mask = np.zeros((y1 - y0, width), dtype=bool)
    for cx, cy, r in ((0.35, 0.4, 0.28), (0.68, 0.62, 0.2)):
        mask |= ((xx - int(cx * width)) ** 2 + (yy - int(cy * height)) ** 2) < int(r * m) ** 2
    band = np.full((y1 - y0, width, 3), 255, dtype=np.uint8)
    if not mask.any():
        return band
    with np.errstate(over="ignore"):                                    # lowbias32 hash of the pixel index
        h = ((yy * width + xx) & 0xFFFFFFFF).astype(np.uint32) + np.uint32((seed * 0x9E3779B1) & 0xFFFFFFFF)
        h ^= h >> np.uint32(16); h *= np.uint32(0x7FEB352D)
        h ^= h >> np.uint32(15); h *= np.uint32(0x846CA68B)
        h ^= h >> np.uint32(16)
    n0 = (h & np.uint32(0xFF)).astype(np.int16) - np.int16(128)
    n1 = ((h >> np.uint32(8)) & np.uint32(0xFF)).astype(np.int16) - np.int16(128)
    # crossed triangle waves stand in for the stroma texture
    tex = (((xx % 44) - 22).astype(np.int16) * ((yy % 56) - 28).astype(np.int16)) // np.int16(32)
    chans = (np.clip(200 + tex + n0 // 10, 0, 255), np.clip(120 + (tex * 3) // 5 + n1 // 12, 0, 255),
             np.clip(170 + (tex * 4) // 5 + n0 // 16, 0, 255))
    for c, ch in enumerate(chans):
        np.copyto(band[:, :, c], ch.astype(np.uint8), where=mask)
    return band


def wsi_slide(width: int, height: int, seed: int = 4, workers: int = 8, band_rows: int = 512) -> np.ndarray:
    """H&E-like RGB slide for the MIC3 configuration sizes (BASELINE.json config 5: 32768 x 32768): white glass with two
    textured, noisy tissue discs (about 37 % of the area), generated band by band on `workers` threads."""
    from concurrent.futures import ThreadPoolExecutor
    out = np.empty((height, width, 3), dtype=np.uint8)

    def fill(y0):
        y1 = min(height, y0 + band_rows)
        out[y0:y1] = wsi_slide_band(width, height, y0, y1, seed)

    with ThreadPoolExecutor(max(1, workers)) as ex:
        list(ex.map(fill, range(0, height, band_rows)))
    return out

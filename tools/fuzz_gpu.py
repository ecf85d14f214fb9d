"""Randomised differential run: GPU (through the C ABI) against the CPU oracle, bit for bit, on many small frames of varied
shape, depth, content and state count.  usage: python tools/fuzz_gpu.py [seconds] [seed].  Exit code 1 on the first mismatch
(the failing case is printed and saved under gpurun_out/)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
mic = entry.load_package()
from oracle import mico

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
LARGE = len(sys.argv) > 3 and sys.argv[3] == "large"          # frames of up to 3000 x 1600: many tiles, many chunks, long runs


def content(w, h, depth):
    mx = (1 << depth) - 1
    kind = rng.integers(0, 9)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0:   img = rng.integers(0, mx + 1, size=(h, w))
    elif kind == 1: img = np.full((h, w), rng.integers(0, mx + 1))
    elif kind == 2: img = (xx * rng.integers(1, 9) + yy * rng.integers(1, 9)) % (mx + 1)
    elif kind == 3: img = (mx / 2 + mx / 3 * np.sin(xx / rng.uniform(2, 40)) * np.cos(yy / rng.uniform(2, 40))).astype(np.int64) + rng.integers(-3, 4, size=(h, w))
    elif kind == 4: img = np.repeat(rng.integers(0, mx + 1, size=(h, (w + 15) // 16)), 16, axis=1)[:, :w]          # long runs
    elif kind == 5:                                                                                              # smooth + spikes (escapes)
        img = np.full((h, w), mx // 2) + rng.integers(-2, 3, size=(h, w)); m = rng.random((h, w)) < 0.02; img[m] = rng.choice([0, mx], size=int(m.sum()))
    elif kind == 6: img = (rng.integers(0, 2, size=(h, w)) * mx)                                                  # two symbols
    elif kind == 7: img = np.cumsum(rng.integers(-1, 2, size=(h, w)), axis=1) + mx // 2
    else:           img = (rng.integers(0, min(mx, 40) + 1, size=(h, w)) + (yy // 7) * 3)
    img = np.clip(img, 0, mx).astype(np.uint16)
    mv = int(img.max()) if rng.random() < 0.7 else mx
    return img, max(mv, 8), kind


def same_rc(a, b):
    """status agreement; the degenerate-RLE case (max value < 8: the reference panics) is ARGS in the oracle
    and UNSUPPORTED in the library"""
    return a == b or {a, b} == {-1, -9}


def fail(what, img, mv, extra):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "fuzz_fail.npy"), img)
    print("MISMATCH", what, img.shape, "max", mv, extra, flush=True)
    sys.exit(1)


t0 = time.time(); n = 0; stats = {}
while time.time() - t0 < budget:
    if LARGE:
        w = int(rng.choice([511, 512, 1000, 2048, 2577, int(rng.integers(300, 3000))]))
        h = int(rng.choice([64, 255, 256, 257, 700, int(rng.integers(60, 1600))]))
    else:
        w = int(rng.choice([1, 2, 3, 7, 16, 63, 64, 65, 127, 200, 257, int(rng.integers(1, 400))]))
        h = int(rng.choice([1, 2, 5, 17, 64, 65, 130, int(rng.integers(1, 300))]))
    depth = int(rng.integers(4, 17))
    img, mv, kind = content(w, h, depth)
    mode = int(rng.integers(0, 9)); ns = int(rng.choice([2, 4, 8]))
    key = ("frame", "grad", "pics", "pica", "mic2", "fse", "wavelet", "wsi", "rgb")[mode]
    try:
        if mode == 0:
            rc, want = mico.compress_single_frame(img, mv, ns)
            got_rc, got = 0, None
            try: got = mic.compress_single_frame(img, w, h, mv, ns)
            except mic.MicError as e: got_rc = e.code
            if not same_rc(rc, got_rc) or (rc == 0 and got != want): fail("frame", img, mv, (ns, rc, got_rc, kind))
            if rc == 0 and not np.array_equal(mic.decompress_single_frame(want, w, h), img): fail("frame decode", img, mv, (ns, kind))
        elif mode == 1:
            rc, want = mico.compress_single_frame_grad(img, mv)
            got_rc, got = 0, None
            try: got = mic.compress_single_frame_grad(img, w, h, mv)
            except mic.MicError as e: got_rc = e.code
            if not same_rc(rc, got_rc) or (rc == 0 and got != want): fail("grad", img, mv, (rc, got_rc, kind))
            if rc == 0 and not np.array_equal(mic.decompress_single_frame_grad(want, w, h), img): fail("grad decode", img, mv, (kind,))
        elif mode == 2:
            strips = int(rng.integers(1, 12))
            rc, want = mico.pics_compress(img, mv, strips, ns)
            got_rc, got = 0, None
            try: got = mic.compress_parallel_strips(img, w, h, mv, strips, ns)
            except mic.MicError as e: got_rc = e.code
            if not same_rc(rc, got_rc) or (rc == 0 and got != want): fail("pics", img, mv, (strips, ns, rc, got_rc, kind))
            if rc == 0 and not np.array_equal(mic.decompress_parallel_strips(want)[0].reshape(h, w), img): fail("pics decode", img, mv, (strips, ns, kind))
        elif mode == 3:
            strips = int(rng.integers(1, 12))
            rc, want = mico.pica_compress(img, mv, strips)
            got_rc, got = 0, None
            try: got = mic.compress_parallel_strips_adaptive(img, w, h, mv, strips)
            except mic.MicError as e: got_rc = e.code
            if not same_rc(rc, got_rc) or (rc == 0 and got != want): fail("pica", img, mv, (strips, rc, got_rc, kind))
            if rc == 0 and not np.array_equal(mic.decompress_parallel_strips_adaptive(want), img): fail("pica decode", img, mv, (strips, kind))
        elif mode == 5:                                                      # bare FSE stage on the frame's token stream or on raw symbols
            sym = mico.delta_rle_compress(img, mv) if rng.random() < 0.5 else img.reshape(-1)
            fl = int(rng.choice([1, 2, 4, 8]))
            rc, want = mico.fse_compress(sym, fl)
            got_rc, got = 0, None
            try: got = mic.fse_compress_u16(sym, fl)
            except mic.MicError as e: got_rc = e.code
            if not same_rc(rc, got_rc) or (rc == 0 and got != want): fail("fse", img, mv, (fl, rc, got_rc, kind))
            if rc == 0 and not np.array_equal(mic.fse_decompress_u16_auto(want, sym.size), sym): fail("fse decode", img, mv, (fl, kind))
        elif mode == 6:
            if w < 2 or h < 2: continue
            lv = int(rng.integers(1, 6))
            rc, want = mico.wavelet_v2_compress(img, mv, lv)
            got_rc, got = 0, None
            try: got = mic.wavelet_v2_compress(img, h, w, mv, lv)
            except mic.MicError as e: got_rc = e.code
            if not same_rc(rc, got_rc) or (rc == 0 and got != want): fail("wavelet", img, mv, (lv, rc, got_rc, kind))
            if rc == 0 and not np.array_equal(mic.wavelet_v2_decompress(want)[0].reshape(h, w), img): fail("wavelet decode", img, mv, (lv, kind))
        elif mode == 7:
            grey = rng.random() < 0.5
            tw, th = int(rng.choice([16, 32, 64, 100])), int(rng.choice([16, 32, 64, 100]))
            if grey:
                src = img if depth > 8 else img.astype(np.uint8)
                rc, want = mico.wsi_compress_grey(src, tw, th, 0)
                got_rc, got = 0, None
                try: got = mic.compress_wsi(src, w, h, channels=1, bits_per_sample=16 if src.dtype == np.uint16 else 8, tile_w=tw, tile_h=th)
                except mic.MicError as e: got_rc = e.code
            else:
                src = np.stack([(img >> k).astype(np.uint8) for k in (0, 2, 4)], axis=-1)
                rc, want = mico.wsi_compress(src, tw, th, 0)
                got_rc, got = 0, None
                try: got = mic.compress_wsi(src, w, h, tile_w=tw, tile_h=th)
                except mic.MicError as e: got_rc = e.code
            if not same_rc(rc, got_rc) or (rc == 0 and got != want): fail("wsi", img, mv, (grey, tw, th, rc, got_rc, kind))
            if rc == 0 and not np.array_equal(mic.decompress_wsi_level(want, 0), src): fail("wsi decode", img, mv, (grey, tw, th, kind))
        elif mode == 8:
            src = np.ascontiguousarray(np.stack([(img >> k).astype(np.uint8) for k in (0, 3, 5)], axis=-1))
            rc, want = mico.wsi_compress_tile(src)
            got_rc, got = 0, None
            try: got = mic.compress_rgb(src, w, h)
            except mic.MicError as e: got_rc = e.code
            if not same_rc(rc, got_rc) or (rc == 0 and got != want): fail("rgb", img, mv, (rc, got_rc, kind))
            if rc == 0 and not np.array_equal(mic.decompress_rgb(want, w, h), src): fail("rgb decode", img, mv, (kind,))
        else:
            nf = int(rng.integers(1, 6)); temporal = bool(rng.integers(0, 2))
            stack = np.stack([np.clip(img.astype(np.int64) + rng.integers(-2, 3, size=img.shape) * (k > 0), 0, 65535).astype(np.uint16) for k in range(nf)])
            smv = max(int(stack.max()), mv)
            rc, want = mico.mic2_compress(stack, smv, temporal)
            got_rc, got = 0, None
            try: got = mic.compress_multi_frame(stack, w, h, smv, temporal=temporal)
            except mic.MicError as e: got_rc = e.code
            if not same_rc(rc, got_rc) or (rc == 0 and got != want): fail("mic2", img, smv, (nf, temporal, rc, got_rc, kind))
            if rc == 0 and not np.array_equal(np.asarray(mic.decompress_multi_frame(want)).reshape(stack.shape), stack): fail("mic2 decode", img, smv, (nf, temporal, kind))
    except SystemExit:
        raise
    stats[(key, rc == 0)] = stats.get((key, rc == 0), 0) + 1
    n += 1
    if n % (10 if LARGE else 200) == 0: print(f"{n} cases, {time.time() - t0:.0f} s", flush=True)
print("OK", n, "cases", sorted(stats.items()))

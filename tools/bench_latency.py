"""Per-call latency of the host-buffer drop-in entry points on ONE XR-shaped frame (what a cgo caller that hands over one
image at a time sees): PICS-8 compress / decompress, wall clock, pageable host memory."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
mic = entry.load_package(); synth = importlib.import_module("medical_image_codec_amd.synth")
W, H = 2577, 2048
img = synth.xr_like(cols=W, rows=H, depth=12, seed=1)
for strips in (8, 16):
    blob = mic.compress_parallel_strips(img, W, H, 4095, strips)
    te, td = [], []
    for _ in range(10):
        t0 = time.perf_counter(); blob = mic.compress_parallel_strips(img, W, H, 4095, strips); te.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); px, w, h = mic.decompress_parallel_strips(blob); td.append(time.perf_counter() - t0)
    assert np.array_equal(np.asarray(px).reshape(img.shape), img)
    e, d = min(te), min(td)
    print(f"PICS-{strips} one {W}x{H} frame: compress {e * 1e3:.2f} ms ({img.nbytes / e / 1e6:.0f} MB/s)  decompress {d * 1e3:.2f} ms ({img.nbytes / d / 1e6:.0f} MB/s)  ratio {img.nbytes / len(blob):.3f}")

"""Run by tests/test_gpu_parity.py::test_sub_batch_loops in a child process with MIC_HIP_WS_BUDGET_MB set small, so that the container
entry points have to cut their unit lists into many sub-batches; every result must still equal the oracle's."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
mic = entry.load_package(); synth = importlib.import_module("medical_image_codec_amd.synth")
from oracle import mico

assert os.environ.get("MIC_HIP_WS_BUDGET_MB"), "meant to run with a small workspace budget"
img = synth.xr_like(cols=300, rows=640, depth=12, seed=17)
for strips, ns in ((16, 2), (10, 4)):
    rc, want = mico.pics_compress(img, 4095, strips, ns)
    assert rc == 0 and mic.compress_parallel_strips(img, 300, 640, 4095, strips, ns) == want
    assert np.array_equal(mic.decompress_parallel_strips(want)[0].reshape(640, 300), img)
rc, want = mico.pica_compress(img, 4095, 4)
assert rc == 0 and mic.compress_parallel_strips_adaptive(img, 300, 640, 4095, 4) == want
assert np.array_equal(mic.decompress_parallel_strips_adaptive(want), img)
rc, _ = mico.pica_compress(img, 4095, 12)                 # 11-row strips at the noisy border: neither predictor's stream normalises
try:
    mic.compress_parallel_strips_adaptive(img, 300, 640, 4095, 12)
    got = 0
except mic.MicError as e:
    got = e.code
assert rc != 0 and got == rc, (rc, got)
stack = np.stack([np.roll(img[:64], 3 * k, axis=1) for k in range(9)])
rc, want = mico.mic2_compress(stack, 4095, False)
assert rc == 0 and mic.compress_multi_frame(stack, 300, 64, 4095) == want
assert np.array_equal(np.asarray(mic.decompress_multi_frame(want)).reshape(stack.shape), stack)
rc, want = mico.mic2_compress(stack, 4095, True)          # temporal: a residual needs the frame before it across sub-batch edges
assert rc == 0 and mic.compress_multi_frame(stack, 300, 64, 4095, temporal=True) == want
assert np.array_equal(np.asarray(mic.decompress_multi_frame(want)).reshape(stack.shape), stack)
assert np.array_equal(np.asarray(mic.decompress_frame(want, 7)).reshape(64, 300), stack[7])
res = mic.compress_batch([stack[k] for k in range(9)], [4095] * 9, 2)
for k, (st, blob, used) in enumerate(res):
    rc, w1 = mico.compress_single_frame(stack[k], 4095, 2)
    assert st == rc == 0 and blob == w1
back = mic.decompress_batch([b for _, b, _ in res], [(300, 64)] * 9)
assert all(st == 0 and np.array_equal(px, stack[k]) for k, (st, px) in enumerate(back))
slide = synth.wsi_like(600, 420, seed=5)                  # 6 + 2 + 1 tiles of 256 x 256: one tile per slab under this ceiling
rc, want = mico.wsi_compress(slide)
assert rc == 0 and mic.compress_wsi(slide, 600, 420) == want
assert np.array_equal(mic.decompress_wsi_level(want, 0), slide)
assert np.array_equal(mic.decompress_wsi_region(want, 0, 130, 120, 400, 280), slide[120:400, 130:530])
frames = np.stack([np.roll(img[:120, :150], 5 * k, axis=0) for k in range(7)])
res = mic.wavelet_v2_compress_batch(frames, 4095, 3)
files = []
for k, (st, blob) in enumerate(res):
    rc, w1 = mico.wavelet_v2_compress(frames[k], 4095, 3)
    assert st == rc == 0 and blob == w1
    files.append(w1)
sts, back = mic.wavelet_v2_decompress_batch(files)
assert sts == [0] * 7 and np.array_equal(back, frames)
print("sub-batch loops ok")

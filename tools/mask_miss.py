"""Diagnostic: what a stale launch mask costs -- a session that has only seen 2-state XR batches meets a 4-state one (its classes are
not in the learned masks: the catch-all kernels take the whole batch), then the same batch again (masks learned)."""
import os, sys, time, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
mic = entry.load_package()
synth = importlib.import_module("medical_image_codec_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 288
W, H, S = 2577, 2048, 8
d_px = synth.xr_like_batch_torch(B, cols=W, rows=H, depth=12, seed0=1, noise=synth.XR_NOISE_PUBLISHED_RATIO, device=torch.device("cuda:0"))
d_out = torch.empty_like(d_px)
sh = (H + S - 1) // S
def units(ns): return mic.Session.make_units([(b * W * H + y0 * W, W, min(H, y0 + sh) - y0, 4095, ns) for b in range(B) for y0 in range(0, H, sh)])
sess = mic.Session(B * S, W * sh)
def step(cu, tag):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sess.encode_enqueue(d_px.data_ptr(), cu); d_blobs, offs, st, ns = sess.encode_finish()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sess.decode_enqueue(d_blobs, offs, cu, d_out.data_ptr()); dst = sess.decode_finish()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    assert (st == 0).all() and (dst == 0).all() and torch.equal(d_out, d_px)
    print(f"{tag}: encode {1e3 * (t1 - t0):8.2f} ms   decode {1e3 * (t2 - t1):8.2f} ms")
u2, u4 = units(2), units(4)
for i in range(3): step(u2, f"2-state #{i}")
for i in range(3): step(u4, f"4-state #{i}")
step(u2, "2-state again")

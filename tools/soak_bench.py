"""Soak: the bench batch (288 XR frames, 2304 strips) encoded and decoded N times in one session; every step's streams, offsets and
pixels compared with the first step's on the device.  usage: python tools/soak_bench.py [steps]"""
import os, sys, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
mic = entry.load_package()
synth = importlib.import_module("medical_image_codec_amd.synth")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, W, H, S = 288, 2577, 2048, 8
dev = torch.device("cuda:0")
d_px = synth.xr_like_batch_torch(B, cols=W, rows=H, depth=12, seed0=1, noise=synth.XR_NOISE_PUBLISHED_RATIO, device=dev)
d_out = torch.empty_like(d_px)
sh = (H + S - 1) // S
cu = mic.Session.make_units([(b * W * H + y0 * W, W, min(H, y0 + sh) - y0, 4095, 2) for b in range(B) for y0 in range(0, H, sh)])
sess = mic.Session(B * S, W * sh)
first = offs0 = None
bad = 0
for it in range(N):
    sess.encode_enqueue(d_px.data_ptr(), cu); d_blobs, offs, st, ns = sess.encode_finish()
    assert (st == 0).all()
    n = int(offs[-1])
    cur = torch.empty(n, dtype=torch.uint8, device=dev); mic.device_copy(cur.data_ptr(), d_blobs, n)
    if first is None: first, offs0 = cur, offs.copy()
    elif not (np.array_equal(offs, offs0) and bool(torch.equal(cur, first))): bad += 1; print("step", it, "streams differ", flush=True)
    d_out.zero_()
    sess.decode_enqueue(d_blobs, offs, cu, d_out.data_ptr()); dst = sess.decode_finish()
    if not ((dst == 0).all() and bool(torch.equal(d_out, d_px))): bad += 1; print("step", it, "pixels differ", flush=True)
    if it % 50 == 49: print("step", it + 1, "ok so far" if not bad else f"{bad} bad", flush=True)
print("soak:", N, "steps,", bad, "bad")
sys.exit(1 if bad else 0)

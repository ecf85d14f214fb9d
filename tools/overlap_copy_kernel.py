"""Diagnostic: does a host<->device copy on its own stream make progress while the decode chain runs?  (The host-pointer pipeline of
mic_host_io.hip counts on it.)  Times a 1 GiB pinned H2D and D2H alone and beside a 288-frame decode, with torch streams and with
the library's own transfer engine (mic_hip_host_alloc buffers, mic_hip_device_copy is not involved)."""
import os, sys, time, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
mic = entry.load_package(); synth = importlib.import_module("medical_image_codec_amd.synth")
W, H, B = 2577, 2048, 288
d_px = synth.xr_like_batch_torch(B, cols=W, rows=H, depth=12, seed0=1, noise=synth.XR_NOISE_PUBLISHED_RATIO)
units = [(b * W * H + y0 * W, W, 256, 4095, 2) for b in range(B) for y0 in range(0, H, 256)]
sess = mic.Session(len(units), W * 256); cu = mic.Session.make_units(units)
sess.encode_enqueue(d_px.data_ptr(), cu); d_blobs, offs, st, ns = sess.encode_finish()
d_out = torch.empty_like(d_px)
host = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
dev = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
def copy(h2d):
    with torch.cuda.stream(side):
        if h2d: dev.copy_(host, non_blocking=True)
        else: host.copy_(dev, non_blocking=True)
def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
def decode():
    sess.decode_enqueue(d_blobs, offs, cu, d_out.data_ptr()); sess.decode_finish()
for h2d in (True, False):
    name = "H2D" if h2d else "D2H"
    for _ in range(2): a = timed(lambda: copy(h2d))
    for _ in range(2): b = timed(decode)
    def both():
        copy(h2d); decode(); side.synchronize()
    for _ in range(2): c = timed(both)
    def both2():
        sess.decode_enqueue(d_blobs, offs, cu, d_out.data_ptr()); copy(h2d); sess.decode_finish(); side.synchronize()
    for _ in range(2): d = timed(both2)
    print(f"{name}: copy alone {a:.1f} ms, decode alone {b:.1f} ms, copy then decode enqueued {c:.1f} ms, decode enqueued then copy {d:.1f} ms")

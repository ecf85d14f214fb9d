"""Diagnostic (build with -DRF_STATS: tools/variants.sh build mic_decode_fused.hip st="-DRF_STATS"): rows by path and shader-clock
ticks by phase of k_dec_rows_tok on the bench workload."""
import os, sys, importlib, ctypes as C
import numpy as np
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
units = [(b * W * H + y0 * W, W, min(H, y0 + sh) - y0, 4095, 2) for b in range(B) for y0 in range(0, H, sh)]
sess = mic.Session(len(units), W * sh)
cu = mic.Session.make_units(units)
sess.encode_enqueue(d_px.data_ptr(), cu)
d_blobs, offs, st, ns = sess.encode_finish()
for it in range(2):
    sess.set_timing(True)
    sess.decode_enqueue(d_blobs, offs, cu, d_out.data_ptr())
    t = sess.last_timings()
    dst = sess.decode_finish()
print("equal:", bool(torch.equal(d_out, d_px)), {k: round(v, 3) for k, v in t if v > 0.03})
L = mic.lib(); out = (C.c_uint32 * 32)()
L.mic_hip_debug_unit.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
tot = np.zeros(8, dtype=np.int64)
for i in range(len(units)):
    L.mic_hip_debug_unit(sess._h, i, out)
    tot += np.array([out[16 + k] for k in range(8)], dtype=np.int64)
n = len(units)
print("per unit: rows %.1f, [ticks: chunk reads + delimiter test %.1f, escapes + e to LDS %.1f], wrapped rows %.1f" % (tot[0] / n, tot[1] / n / 1e3, tot[2] / n / 1e3, tot[3] / n))
print("per unit, 1e3 ticks (x16): assemble %.1f, issue %.1f, predictor %.1f, put %.1f" % tuple(tot[4:] / n / 1e3))
for i in (0, 7, 1000):
    L.mic_hip_debug_unit(sess._h, i, out)
    print("unit", i, [out[16 + k] for k in range(8)])

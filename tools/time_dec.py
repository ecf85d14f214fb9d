"""Timing of the decode kernels on the bench workload, ignoring decode status (for ablation builds whose output is wrong).
usage: python tools/time_dec.py [frames] [nstates]"""
import os, sys, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
mic = entry.load_package()
synth = importlib.import_module("medical_image_codec_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 288
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W, H, S = 2577, 2048, 8
base = [synth.xr_like(cols=W, rows=H, depth=12, seed=1 + i) for i in range(4)]
host = np.stack([base[i % 4] for i in range(B)])
d_px = torch.from_numpy(host.view(np.int16)).cuda()
d_out = torch.empty_like(d_px)
sh = (H + S - 1) // S
units = [(b * W * H + y0 * W, W, min(H, y0 + sh) - y0, 4095, NS) for b in range(B) for y0 in range(0, H, sh)]
sess = mic.Session(len(units), W * sh)
cu = mic.Session.make_units(units)
sess.encode_enqueue(d_px.data_ptr(), cu)
d_blobs, offs, st, ns = sess.encode_finish()
print("encode ok", (st == 0).all(), "ratio", host.nbytes / int(offs[-1]))
for it in range(3):
    sess.set_timing(True)
    sess.decode_enqueue(d_blobs, offs, cu, d_out.data_ptr())
    t = sess.last_timings()
    dst = sess.decode_finish()
    if it == 2:
        print("decode status ok:", int((dst == 0).sum()), "of", len(units), " equal:", bool(torch.equal(d_out, d_px)))
        print({k: round(v, 3) for k, v in t if v > 0.03})

import ctypes as C
L = mic.lib()
out = (C.c_uint32 * 32)()
L.mic_hip_debug_unit.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
for i in (0, 9, 900):
    L.mic_hip_debug_unit(sess._h, i, out)
    d = [out[16 + k] * 16 for k in range(5)]
    if d[4]:
        print("unit", i, "cycles per chunk: rounds %.0f  ring stores %.0f  loads %.0f  state stores %.0f  (chunks %d)" % tuple([d[k] / (d[4] / 16) for k in range(4)] + [d[4] // 16]))

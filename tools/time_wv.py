"""WaveletV2 decode timing of a few CR-shaped frames with the chain kernel's per-phase stamps (LS_STAMP build).
usage: python tools/time_wv.py [frames]"""
import os, sys, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
import ctypes as C
mic = entry.load_package()
synth = importlib.import_module("medical_image_codec_amd.synth")
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows, cols = 2140, 1760
d_px = synth.xr_like_batch_torch(nf, cols=cols, rows=rows, depth=12, seed0=2000, noise=5.0, device="cuda")
d_out = torch.empty_like(d_px)
sess = mic.Session(nf, 2 * rows * cols + 16)
d_s, offs, st, ap = sess.wavelet_v2_encode(d_px.data_ptr(), nf, rows, cols, 5)
for it in range(3):
    sess.set_timing(1)
    dst = sess.wavelet_v2_decode(d_s, offs, nf, rows, cols, ap, d_out.data_ptr())
    t = sess.last_timings()
print("status ok", int((dst == 0).sum()), "of", nf, "equal", bool(torch.equal(d_out, d_px)))
print({k: round(v, 3) for k, v in t if v > 0.03})
L = mic.lib()
out = (C.c_uint32 * 32)()
L.mic_hip_debug_unit.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
for i in range(min(nf, 3)):
    L.mic_hip_debug_unit(sess._h, i, out)
    print('unit', i, 'ntok', out[0], 'nseg', out[12], 'nsym', out[13], 'dbg8..15', [out[16 + k] for k in range(8, 16)])
    d = [out[16 + k] * 16 for k in range(5)]
    if d[4]:
        print("unit", i, "cycles per chunk: rounds %.0f  ring stores %.0f  loads %.0f  state stores %.0f  (chunks %d)" % tuple([d[k] / (d[4] / 16) for k in range(4)] + [d[4] // 16]))

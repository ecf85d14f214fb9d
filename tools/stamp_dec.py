"""Diagnostic (build with EXTRA_FLAGS=-DMIC_STAMP): s_memtime ticks (100 MHz) per phase of k_dec_pixels_wg."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import torch
mic = entry.load_package(); synth = importlib.import_module("medical_image_codec_amd.synth")
W, H = 2577, 2048
img = synth.xr_like(cols=W, rows=H, depth=12, seed=1)
d_px = torch.from_numpy(img.view(np.int16)).cuda(); d_out = torch.empty_like(d_px)
units = [(y0 * W, W, 256, 4095, int(os.environ.get("NS", "2"))) for y0 in range(0, H, 256)]
sess = mic.Session(len(units), W * 256); cu = mic.Session.make_units(units)
sess.encode_enqueue(d_px.data_ptr(), cu); d_blobs, offs, st, ns = sess.encode_finish(); assert (st == 0).all()
sess.decode_enqueue(d_blobs, offs, cu, d_out.data_ptr()); assert (sess.decode_finish() == 0).all()
assert torch.equal(d_out, d_px)
buf = (C.c_uint32 * 16)()
for i in range(len(units)):
    mic.lib().mic_hip_debug_unit(sess._h, i, buf)
    print(f"unit {i}: ntok={buf[0]} nseg={buf[12]} nsym={buf[13]} ticks: walk={buf[4]} expand={buf[5]} scan={buf[6]} wavefront={buf[7]}")

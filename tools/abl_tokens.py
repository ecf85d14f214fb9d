"""Timing-only ablation helper: encode the bench batch (288 distinct frames at the published-ratio noise, as bench.py makes them) and print the tokeniser's kernel time, ignoring statuses
(for builds whose -DTK_ABL_* switches make the output invalid on purpose)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import torch
mic = entry.load_package(); synth = importlib.import_module("medical_image_codec_amd.synth")
W, H, B = 2577, 2048, int(os.environ.get("FRAMES", "288"))
d_px = synth.xr_like_batch_torch(B, cols=W, rows=H, depth=12, seed0=1, noise=synth.XR_NOISE_PUBLISHED_RATIO)   # the bench workload: distinct frames
units = [(b * W * H + y0 * W, W, 256, 4095, 2) for b in range(B) for y0 in range(0, H, 256)]
sess = mic.Session(len(units), W * 256); cu = mic.Session.make_units(units)
for rep in range(3):
    sess.set_timing(True)
    sess.encode_enqueue(d_px.data_ptr(), cu); te = sess.last_timings()
    sess.encode_finish()
print({k: round(v, 3) for k, v in te if v > 0.1})

"""Where the host side of a bench step goes: wall time of each of the four session calls (enqueue / finish, encode / decode) on the
bench batch, next to the kernel time of the step.  usage: python tools/host_gaps.py [frames]"""
import os, sys, time, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
mic = entry.load_package()
synth = importlib.import_module("medical_image_codec_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 288
W, H, S = 2577, 2048, 8
base = [synth.xr_like(cols=W, rows=H, depth=12, seed=1 + i) for i in range(4)]
d_px = torch.from_numpy(np.stack([base[i % 4] for i in range(B)]).view(np.int16)).cuda()
d_out = torch.empty_like(d_px)
sh = (H + S - 1) // S
units = [(b * W * H + y0 * W, W, min(H, y0 + sh) - y0, 4095, 2) for b in range(B) for y0 in range(0, H, sh)]
sess = mic.Session(len(units), W * sh)
cu = mic.Session.make_units(units)
def step(rec):
    t = [time.perf_counter()]
    sess.encode_enqueue(d_px.data_ptr(), cu); t.append(time.perf_counter())
    d_blobs, offs, st, ns = sess.encode_finish(); t.append(time.perf_counter())
    sess.decode_enqueue(d_blobs, offs, cu, d_out.data_ptr()); t.append(time.perf_counter())
    sess.decode_finish(); t.append(time.perf_counter())
    if rec is not None: rec.append([1e3 * (b - a) for a, b in zip(t, t[1:])])
for _ in range(3): step(None)
rec = []
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step(rec)
torch.cuda.synchronize()
tot = 1e3 * (time.perf_counter() - t0) / 10
m = np.median(np.array(rec), axis=0)
print("ms per step %.3f | encode_enqueue returns after %.3f, encode_finish %.3f, decode_enqueue %.3f, decode_finish %.3f (medians)" % (tot, *m))

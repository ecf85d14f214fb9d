"""Per-kernel milliseconds of one PICS-8 encode + decode of 256 XR-shaped frames for each state count (device-resident session)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import torch
mic = entry.load_package(); synth = importlib.import_module("medical_image_codec_amd.synth")
W, H, F = 2577, 2048, int(os.environ.get("FRAMES", "256"))
img = synth.xr_like(cols=W, rows=H, depth=12, seed=1)
host = np.stack([img] * F)
d_px = torch.from_numpy(host.view(np.int16)).cuda(); d_out = torch.empty_like(d_px)
for ns in (2, 4, 8):
    units = [(f * W * H + y0 * W, W, 256, 4095, ns) for f in range(F) for y0 in range(0, H, 256)]
    sess = mic.Session(len(units), W * 256); cu = mic.Session.make_units(units)
    for rep in range(2):
        sess.set_timing(True)
        sess.encode_enqueue(d_px.data_ptr(), cu); te = sess.last_timings()
        d_blobs, offs, st, used = sess.encode_finish(); assert (st == 0).all()
        sess.decode_enqueue(d_blobs, offs, cu, d_out.data_ptr()); td = sess.last_timings()
        assert (sess.decode_finish() == 0).all()
    assert torch.equal(d_out, d_px)
    enc = sum(ms for _, ms in te); dec = sum(ms for _, ms in td)
    big = ", ".join(f"{n} {ms:.2f}" for n, ms in te + td if ms > 1.0)
    print(f"{ns}-state: encode {enc:.1f} ms, decode {dec:.1f} ms, {host.nbytes / (enc + dec) / 1e6:.1f} GB/s  ({big})")
    sess.close()

"""Diagnostic (build with EXTRA_FLAGS=-DMIC_STAMP): shader-clock ticks per phase of k_dec_pixels_wg, lone (8 strips) and
inside a full batch (FRAMES x 8 strips)."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import torch
mic = entry.load_package(); synth = importlib.import_module("medical_image_codec_amd.synth")
W, H = 2577, 2048
img = synth.xr_like(cols=W, rows=H, depth=12, seed=1)
for F in (1, 128, int(os.environ.get("FRAMES", "256"))):
    host = np.stack([img] * F)
    d_px = torch.from_numpy(host.view(np.int16)).cuda(); d_out = torch.empty_like(d_px)
    units = [(f * W * H + y0 * W, W, 256, 4095, 2) for f in range(F) for y0 in range(0, H, 256)]
    sess = mic.Session(len(units), W * 256); cu = mic.Session.make_units(units)
    sess.encode_enqueue(d_px.data_ptr(), cu); d_blobs, offs, st, ns = sess.encode_finish(); assert (st == 0).all()
    sess.decode_enqueue(d_blobs, offs, cu, d_out.data_ptr()); assert (sess.decode_finish() == 0).all()
    buf = (C.c_uint32 * 32)()
    for i in (0, len(units) // 2, len(units) - 1):
        mic.lib().mic_hip_debug_unit(sess._h, i, buf)
        clk = buf[24] / max(buf[25], 1) * 100.0
        print(f"F={F} unit {i}: px.fetch={buf[16]} px.scan={buf[17]} px.number={buf[18]} px.store={buf[19]}  tans: memtime={buf[24]} realtime={buf[25]} -> {clk:.0f} MHz, {buf[24] / max(buf[26] * 64, 1):.1f} ticks/pair over {buf[26]} chunks of the longer half; prologue {buf[27]} ticks")
    t = []
    for i in range(len(units)):
        mic.lib().mic_hip_debug_unit(sess._h, i, buf)
        t.append(buf[25] / 100.0)                                 # microseconds inside the chunk loop + tails (s_memrealtime, 100 MHz)
    t = np.array(t)
    slow = np.nonzero(t > np.percentile(t, 90) * 1.05)[0]
    print(f"F={F}: {slow.size} slow units:", slow[:64].tolist())
    print(f"F={F}: tANS loop time per unit, us: min {t.min():.0f} median {np.median(t):.0f} p90 {np.percentile(t, 90):.0f} p99 {np.percentile(t, 99):.0f} max {t.max():.0f}")
    sess.close(); del d_px, d_out

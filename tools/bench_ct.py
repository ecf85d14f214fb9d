"""CT-shaped check: 16-bit 512x512 frames (tableLog 16 alphabets), one unit per frame, device-resident session.
Prints per-kernel milliseconds of one encode and one decode of the batch."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import torch
mic = entry.load_package(); synth = importlib.import_module("medical_image_codec_amd.synth")
F = int(os.environ.get("FRAMES", "256")); S = 512
ct = np.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "CT_512_512_image.bin"), dtype="<u2").reshape(S, S)
host = np.stack([np.roll(ct, i % 8, axis=1) for i in range(F)])       # the reference's CT test image, shifted copies
d_px = torch.from_numpy(host.view(np.int16)).cuda(); d_out = torch.empty_like(d_px)
mv = int(host.max())
units = [(i * S * S, S, S, mv, 2) for i in range(F)]
sess = mic.Session(F, S * S); cu = mic.Session.make_units(units)
for rep in range(2):
    sess.set_timing(True)
    sess.encode_enqueue(d_px.data_ptr(), cu); te = sess.last_timings()
    d_blobs, offs, st, ns = sess.encode_finish(); assert (st == 0).all(), st[:8]
    sess.decode_enqueue(d_blobs, offs, cu, d_out.data_ptr()); td = sess.last_timings()
    assert (sess.decode_finish() == 0).all()
assert torch.equal(d_out, d_px)
raw = host.nbytes
print(f"{F} frames {S}x{S}, max {mv}, ratio {raw / int(offs[-1]):.3f}")
for name, ms in te + td:
    if ms > 0.05: print(f"  {name:34s} {ms:8.3f} ms")
enc = sum(ms for _, ms in te); dec = sum(ms for _, ms in td)
print(f"  encode {raw / enc / 1e6:.1f} GB/s   decode {raw / dec / 1e6:.1f} GB/s (kernel time)")
if os.environ.get("MIC_STAMP"):
    import ctypes as C
    buf = (C.c_uint32 * 32)()
    sess.encode_enqueue(d_px.data_ptr(), cu); sess.encode_finish()
    mic.lib().mic_hip_debug_unit(sess._h, 0, buf)
    names = ["tok.A", "tok.B", "tok.C", "tok.D", "tok.E", "tab.normalise", "tab.ncount", "tab.ctable", "tans.walk", "tans.fixup", "tans.bits", "tans.pack"]
    print("unit 0 ticks: tl", buf[2], "symlen", buf[3], " ".join(f"{n}={buf[16 + k]}" for k, n in enumerate(names)))

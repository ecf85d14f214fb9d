"""Debug: dump the first rounds of the lane-per-state decoder (LS_DEBUG build) for a 4-state frame."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
mic = entry.load_package()
import torch
from oracle import mico
mico.lib()
img = np.fromfile(os.path.join(ROOT, "tests/golden/MR_256_256_image.bin"), dtype="<u2").reshape(256, 256)
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rc, blob = mico.compress_single_frame(img, int(img.max()), ns)
tok = mico.delta_rle_compress(img, int(img.max()))
print("tokens", tok[:24])
d_blob = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).cuda()
d_out = torch.zeros(256 * 256, dtype=torch.int16, device="cuda")
sess = mic.Session(1, 256 * 256)
units = mic.Session.make_units([(0, 256, 256, int(img.max()), ns)])
offs = np.array([0, len(blob)], dtype=np.uint64)
sess.decode_enqueue(d_blob.data_ptr(), offs, units, d_out.data_ptr())
st = sess.decode_finish()
print("status", st)
buf = np.zeros(16 * 64, dtype=np.uint32)
L = mic.lib()
L.mic_hip_debug_fetch_hist.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
print("fetch", L.mic_hip_debug_fetch_hist(sess._h, 0, buf.ctypes.data, buf.nbytes))
for r in range(7, 13):
    for k in range(ns if ns < 100 else 8):
        d = buf[r * 64 + k * 8: r * 64 + k * 8 + 8]
        print(f"r{r} k{k} st={d[0]} nb={d[1]} pre={d[2]} q={np.int32(d[3])} e={d[4]} hi={d[5]:08x} tot={d[6]} C={d[7]}")

import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import numpy as np, torch
mic = entry.load_package()
synth = importlib.import_module("medical_image_codec_amd.synth")
nf = int(sys.argv[1]); rows, cols = 2140, 1760
d_px = synth.xr_like_batch_torch(nf, cols=cols, rows=rows, depth=12, seed0=2000, noise=5.0, device="cuda")
d_out = torch.empty_like(d_px)
sess = mic.Session(nf, 2 * rows * cols + 16)
d_s, offs, st, ap = sess.wavelet_v2_encode(d_px.data_ptr(), nf, rows, cols, 5)
print("enc", np.unique(st, return_counts=True), ap, offs[:3])
dst = sess.wavelet_v2_decode(d_s, offs, nf, rows, cols, ap, d_out.data_ptr())
print("dec", np.unique(dst, return_counts=True), "first bad", np.nonzero(dst)[0][:10])
eq = (d_out == d_px).reshape(nf, -1).all(dim=1).cpu().numpy()
print("frames equal", int(eq.sum()), "of", nf, "first unequal", np.nonzero(~eq)[0][:10])

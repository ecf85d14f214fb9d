"""Token stream of the library's tokeniser against the oracle's (DeltaRleCompressU16), unit by unit: first mismatch and where it lies."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import torch, importlib
mic = entry.load_package()
from oracle import mico
synth = importlib.import_module("medical_image_codec_amd.synth")
L = mic.lib()
L.mic_hip_debug_fetch_tok.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]

def check(name, img, maxv):
    h, w = img.shape
    d_px = torch.from_numpy(np.ascontiguousarray(img).view(np.int16)).cuda()
    sess = mic.Session(1, w * h)
    cu = mic.Session.make_units([(0, w, h, maxv, 2)])
    sess.encode_enqueue(d_px.data_ptr(), cu); d_blobs, offs, st, used = sess.encode_finish()
    buf = (C.c_uint32 * 32)(); L.mic_hip_debug_unit(sess._h, 0, buf)
    ntok = buf[0]
    want = mico.delta_rle_compress(img, maxv)
    got = np.empty(max(ntok, 1), dtype=np.uint16)
    rc = L.mic_hip_debug_fetch_tok(sess._h, 0, got.ctypes.data, ntok)
    m = min(len(want), ntok)
    bad = np.nonzero(got[:m] != want[:m])[0]
    depth = int(maxv).bit_length(); c = (1 << (depth - 1)) - 1 - 3
    print(f"{name}: {w}x{h} maxv {maxv} c {c} status {st} ntok gpu {ntok} oracle {len(want)} mismatches {len(bad)}", "first", bad[:8] if len(bad) else None)
    if len(bad):
        i = int(bad[0])
        print("   gpu ", got[max(0, i - 6): i + 10]); print("   want", want[max(0, i - 6): i + 10])
    sess.close()

xr = synth.xr_like(cols=2577, rows=2048, depth=12, seed=1, noise=synth.XR_NOISE_PUBLISHED_RATIO)
for s0 in (0, 3):
    check(f"xr strip {s0}", xr[s0 * 256:(s0 + 1) * 256], 4095)
rng = np.random.default_rng(5)
check("noise 12-bit 512x64", (2000 + rng.normal(0, 40, (64, 512))).astype(np.uint16), 4095)
check("noise 8-bit 256x256", (120 + rng.normal(0, 9, (256, 256))).clip(0, 255).astype(np.uint16), 255)
check("noise 8-bit sigma 2", (120 + rng.normal(0, 2, (256, 256))).clip(0, 255).astype(np.uint16), 255)
check("ramp 10-bit", ((np.arange(256 * 300).reshape(300, 256) // 7) % 1000).astype(np.uint16), 1023)
check("esc-heavy 12-bit", rng.integers(0, 4096, (40, 300)).astype(np.uint16), 4095)
check("above-max 12-bit", rng.integers(0, 65536, (33, 129)).astype(np.uint16), 4095)
check("16-bit noise", rng.integers(0, 65536, (64, 257)).astype(np.uint16), 65535)
check("16-bit smooth", (30000 + rng.normal(0, 30, (128, 515))).astype(np.uint16), 65535)
check("w7", (100 + rng.normal(0, 3, (700, 7))).astype(np.uint16), 255)
check("w9", (100 + rng.normal(0, 3, (700, 9))).astype(np.uint16), 255)
check("w8", (100 + rng.normal(0, 3, (1100, 8))).astype(np.uint16), 255)
check("one row", (1000 + rng.normal(0, 30, (1, 9000))).astype(np.uint16), 4095)
check("tiny", np.array([[5, 5, 5], [5, 6, 5]], dtype=np.uint16), 15)
check("esc in half 2", np.concatenate([(1000 + rng.normal(0, 3, (3, 1024))), np.full((1, 1024), 4000.0), (1000 + rng.normal(0, 3, (8, 1024)))]).astype(np.uint16), 4095)
z = (500 + rng.normal(0, 1.0, (64, 4096))).astype(np.uint16); z[:, 1000:3000] = 77
check("long runs", z, 1023)

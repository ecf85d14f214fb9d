"""Wall-clock check of the host-buffer entry points that bench.py does not cover: WaveletV2 on a CR-shaped frame
(BASELINE configs[2]), MIC3 / WSI on a synthetic RGB slide (configs[4], scaled down), MIC2 on a CT-shaped stack (configs[3]).
Host buffers in and out, so PCIe and host-side container assembly are inside these numbers."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
mic = entry.load_package(); synth = importlib.import_module("medical_image_codec_amd.synth")


def timed(f, reps=3):
    f(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0)
    return r, best


cr = synth.cr_like()
rows, cols = cr.shape
blob, te = timed(lambda: mic.wavelet_v2_compress(cr, rows, cols, 4095, 5))
(px, r, c), td = timed(lambda: mic.wavelet_v2_decompress(blob))
assert np.array_equal(np.asarray(px).reshape(cr.shape), cr)
print(f"WaveletV2 CR {cols}x{rows}: ratio {cr.nbytes / len(blob):.3f}  encode {cr.nbytes / te / 1e6:.0f} MB/s  decode {cr.nbytes / td / 1e6:.0f} MB/s")

NB = int(os.environ.get("WV_FRAMES", "48"))
stack = np.stack([np.roll(cr, 7 * k, axis=1) for k in range(NB)])
res, te = timed(lambda: mic.wavelet_v2_compress_batch(stack, 4095, 5), reps=2)
assert all(st == 0 for st, _ in res) and res[0][1] == blob
files = [b for _, b in res]
(sts, back), td = timed(lambda: mic.wavelet_v2_decompress_batch(files), reps=2)
assert sts == [0] * NB and np.array_equal(back, stack)
print(f"WaveletV2 CR batch of {NB}: encode {stack.nbytes / te / 1e6:.0f} MB/s  decode {stack.nbytes / td / 1e6:.0f} MB/s  (host buffers, wall clock)")

S = int(os.environ.get("SLIDE", "8192"))
slide = synth.wsi_like(S, S, seed=4)
blob, te = timed(lambda: mic.compress_wsi(slide, S, S), reps=2)
lv0, td = timed(lambda: mic.decompress_wsi_level(blob, 0), reps=2)
assert np.array_equal(np.asarray(lv0).reshape(slide.shape), slide)
print(f"MIC3 RGB {S}x{S}: ratio {slide.nbytes / len(blob):.2f}  encode {slide.nbytes / te / 1e6:.0f} MB/s  decode(level 0) {slide.nbytes / td / 1e6:.0f} MB/s")

F = int(os.environ.get("FRAMES", "128"))
ct = np.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "CT_512_512_image.bin"), dtype="<u2").reshape(512, 512)
stack = np.stack([np.roll(ct, i % 16, axis=1) for i in range(F)])
mv = int(stack.max())
blob, te = timed(lambda: mic.compress_multi_frame(stack, 512, 512, mv), reps=2)
back, td = timed(lambda: mic.decompress_multi_frame(blob), reps=2)
assert np.array_equal(back, stack)
print(f"MIC2 {F} x 512x512 CT: ratio {stack.nbytes / len(blob):.3f}  encode {stack.nbytes / te / 1e6:.0f} MB/s  decode {stack.nbytes / td / 1e6:.0f} MB/s")

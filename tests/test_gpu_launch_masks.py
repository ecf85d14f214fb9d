"""Launch masks (csrc/mic_launch.h): a session launches only the kernel classes its last batches used, and the catch-all kernels take
what a stale mask leaves.  One session meets, batch after batch, streams of classes it has not seen -- other state counts, other table
sizes, small and large units, other frame widths -- and every stream must equal the oracle's, every frame must come back, on the
FIRST batch of a new kind (the mask is stale) and on the next (it has learned)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _round_trip(mic, mico, sess, torch, imgs, maxv, ns):
    h, w = imgs[0].shape
    units = mic.Session.make_units([(i * w * h, w, h, maxv, ns) for i in range(len(imgs))])
    d_px = torch.from_numpy(np.stack(imgs).view(np.int16).copy()).cuda()
    sess.encode_enqueue(d_px.data_ptr(), units)
    d_blobs, offs, st, used = sess.encode_finish()
    host = np.empty(int(offs[-1]), np.uint8)
    assert C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(d_blobs), C.c_size_t(host.size), 2) == 0
    ok = []
    for k, img in enumerate(imgs):
        rc, want = mico.compress_single_frame(img, maxv, ns)
        assert st[k] == rc, (k, st[k], rc)
        if rc == 0:
            assert host[int(offs[k]):int(offs[k + 1])].tobytes() == want, (k, ns)
            ok.append(k)
    d_out = torch.zeros(len(imgs) * w * h, dtype=torch.int16, device="cuda")
    sess.decode_enqueue(d_blobs, offs, units, d_out.data_ptr())
    dst = sess.decode_finish()
    back = d_out.cpu().numpy().view(np.uint16).reshape(len(imgs), h, w)
    for k in ok:
        assert dst[k] == 0 and np.array_equal(back[k], imgs[k]), (k, ns)


def test_a_session_meets_classes_its_masks_have_not_seen(mic, mico, synth, gpu_ready):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(5)
    kinds = [
        # (frames, max value, states): what is new about the batch
        ([synth.xr_like(cols=700, rows=600, depth=12, seed=i) for i in range(3)], 4095, 2),          # tableLog 13, two states
        ([synth.xr_like(cols=700, rows=600, depth=12, seed=i) for i in range(3)], 4095, 4),          # four states (wide encoder, another decode class)
        ([synth.xr_like(cols=700, rows=600, depth=12, seed=9 + i) for i in range(3)], 4095, 8),      # eight
        ([synth.xr_like(cols=700, rows=600, depth=8, seed=20 + i, noise=2.0) for i in range(3)], 255, 2),   # small alphabet: tableLog <= 12
        ([synth.xr_like(cols=700, rows=600, depth=16, seed=30 + i) for i in range(3)], 65535, 2),    # deep: tableLog 14-16, tier 2
        ([synth.xr_like(cols=700, rows=600, depth=12, seed=40 + i) for i in range(3)], 4095, 2),     # and back
        ([rng.integers(0, 4096, (600, 700), dtype=np.uint16) for _ in range(2)], 4095, 2),           # noise: the two-state attempt is handed on / fails
    ]
    sess = mic.Session(3, 700 * 600)
    try:
        for imgs, maxv, ns in kinds:
            for again in range(2):                                                # stale mask, then learned
                _round_trip(mic, mico, sess, torch, imgs, maxv, ns)
    finally:
        sess.close()
    # other widths through one session: the predictor classes come from the host (widths), the entropy classes from memory
    sess = mic.Session(2, 2577 * 64)
    try:
        for w in (2577, 1100, 640, 3000, 2577):
            _round_trip(mic, mico, sess, torch, [synth.xr_like(cols=w, rows=64, depth=12, seed=w + i) for i in range(2)], 4095, 2)
    finally:
        sess.close()

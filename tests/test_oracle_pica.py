"""PICA -- content-adaptive strips with a per-strip predictor choice (parallelstripsadaptive.go) and the gradient-adaptive
predictor behind it (deltagradrlecompressu16.go, deltagradcompressu16.go).  The reference holds no golden vectors for this
path (its tests are round trips, parallelstripsadaptive_test.go:14-44, deltagradcompressu16_test.go:12-60); the oracle is
pinned by the reference's PUBLISHED results on its own test images (docs/adaptive-compression.md:30-41, :70-83: ratios to four
digits and the number of strips that chose the gradient predictor), checked against an independent numpy / pure-Python
restatement of the two pieces that are new -- the predictor and the float64 boundary search -- and through round trips; the
FSE / RLE stages underneath are the byte-pinned ones."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def grad_predict_np(w, n, nw, ne):
    """gradPredict, deltagradcompressu16.go:147-167, vectorised"""
    w, n, nw, ne = (a.astype(np.int64) for a in (w, n, nw, ne))
    avg = (w + n) >> 1
    g = np.abs(w - nw) + np.abs(n - nw)
    corr = np.clip((ne - nw) >> 3, -(g >> 1), g >> 1)
    return np.where(g == 0, avg, avg + corr)


def grad_symbols_np(px, max_value):
    """the escape / threshold symbols of GradDeltaCompressU16 (deltagradcompressu16.go:20-62), before RLE"""
    h, w = px.shape
    depth = int(max_value).bit_length()
    thr, delim = (1 << (depth - 1)) - 1, (1 << depth) - 1
    p = px.astype(np.int64)
    pred = np.zeros_like(p)
    pred[0, 1:] = p[0, :-1]
    pred[1:, 0] = p[:-1, 0]
    ne = np.concatenate([p[:-1, 2:], p[:-1, -2:-1]], axis=1) if w > 1 else p[:-1, 1:]
    if w > 1:
        pred[1:, 1:] = grad_predict_np(p[1:, :-1], p[:-1, 1:], p[:-1, :-1], ne)
    diff = p - pred
    out = [int(max_value)]
    for d, v in zip(diff.reshape(-1).tolist(), p.reshape(-1).tolist()):
        if (abs(d) & 0xFFFF) >= thr:
            out += [delim, v]
        else:
            out.append(thr + d)
    return np.array(out, dtype=np.uint16)


def boundaries_py(px, num_strips):
    """adaptiveStripBoundaries, parallelstripsadaptive.go:222-289, with Python floats (IEEE doubles, like Go's float64)"""
    h, w = px.shape
    if num_strips >= h:
        return list(range(h))
    if num_strips == 1:
        return [0]
    p = px.astype(np.int64)
    cost = [0.0] + [float(int(np.abs(p[y] - p[y - 1]).sum())) for y in range(1, h)]
    cum = [0.0]
    for c in cost:
        cum.append(cum[-1] + c)
    total = cum[h]
    starts = [0]
    if total == 0:
        return starts + [i * h // num_strips for i in range(1, num_strips)]
    for i in range(1, num_strips):
        target = total * float(i) / float(num_strips)
        lo, hi = starts[-1] + 1, h
        while lo < hi:
            mid = (lo + hi) >> 1
            if cum[mid] < target:
                lo = mid + 1
            else:
                hi = mid
        starts.append(min(lo, h - 1))
    return starts


def _images(synth):
    mr = np.fromfile(os.path.join(GOLDEN, "MR_256_256_image.bin"), dtype="<u2").reshape(256, 256)
    ct = np.fromfile(os.path.join(GOLDEN, "CT_512_512_image.bin"), dtype="<u2").reshape(512, 512)
    xr = synth.xr_like(cols=601, rows=403, depth=12, seed=4)
    return [("MR", mr, int(mr.max())), ("CT", ct, int(ct.max())), ("XR", xr, 4095)]


def rle_expand_py(tok):
    """RleDecompressU16.DecodeNext2 (rledecompressu16.go:59-85) run to the end of the token stream"""
    tok = [int(t) for t in tok]
    mid = (1 << (tok[0].bit_length() - 1)) - 1
    out, i = [], 1
    while i < len(tok):
        c = tok[i]; i += 1
        if c <= mid:
            out += [tok[i]] * c; i += 1
        else:
            out += tok[i: i + c - mid]; i += c - mid
    return np.array(out, dtype=np.uint16)


def test_grad_symbols_match_the_numpy_restatement(mico, synth):
    for name, img, mx in _images(synth):
        sub = img[:64, :97] if name != "XR" else img[:50, :]
        want = grad_symbols_np(sub, mx)
        tok = mico.grad_delta_rle_compress(sub, mx)
        assert np.array_equal(rle_expand_py(tok), want), name
        rc, back = mico.grad_delta_rle_decompress(tok, sub.shape[1], sub.shape[0])
        assert rc == 0 and np.array_equal(back, sub), name
        assert not np.array_equal(mico.delta_rle_compress(sub, mx), tok), name      # the avg predictor gives another stream


def test_grad_edge_cases(mico):
    rng = np.random.default_rng(5)
    for w, h in ((1, 1), (1, 7), (9, 1), (2, 2), (3, 5), (64, 3)):
        px = rng.integers(0, 4096, size=(h, w), dtype=np.uint16)
        tok = mico.grad_delta_rle_compress(px, 4095)
        rc, back = mico.grad_delta_rle_decompress(tok, w, h)
        assert rc == 0 and np.array_equal(back, px), (w, h)
    px = np.zeros((16, 16), np.uint16); px[::2] = 65535                      # every delta escapes at 16 bits
    tok = mico.grad_delta_rle_compress(px, 65535)
    rc, back = mico.grad_delta_rle_decompress(tok, 16, 16)
    assert rc == 0 and np.array_equal(back, px)


def test_single_frame_grad_round_trip_and_gain(mico, synth):
    for name, img, mx in _images(synth):
        rc, blob = mico.compress_single_frame_grad(img, mx)
        assert rc == 0
        rc, back = mico.decompress_single_frame_grad(blob, img.shape[1], img.shape[0])
        assert rc == 0 and np.array_equal(back, img), name
        rc, avg = mico.compress_single_frame(img, mx, 2)
        assert rc == 0
        if name == "CT":                                                       # deltagradcompressu16.go:143-145: CT is the one regression
            assert len(blob) > len(avg)
        if name == "MR":
            assert len(blob) < len(avg)


@pytest.mark.parametrize("strips", [1, 2, 5, 8, 16])
def test_pica_boundaries_match_python_floats(mico, synth, strips):
    for name, img, mx in _images(synth):
        assert mico.pica_boundaries(img, strips) == boundaries_py(img, strips), name
    flat = np.full((40, 16), 7, np.uint16)
    assert mico.pica_boundaries(flat, 4) == [0, 10, 20, 30]                    # uniform image: equal heights (:262-268)
    assert mico.pica_boundaries(flat[:3], 8) == [0, 1, 2]                      # more strips than rows (:223-229)


@pytest.mark.parametrize("strips", [1, 4, 8])
def test_pica_container_and_round_trip(mico, synth, strips):
    for name, img, mx in _images(synth):
        h, w = img.shape
        rc, blob = mico.pica_compress(img, mx, strips)
        if name == "XR" and strips == 8:                                       # the cost partition leaves a 7-row and a 5-row strip at the noisy
            assert rc in (-8, -10)                                             # borders; neither predictor's stream normalises: the reference
            continue                                                           # returns that strip's error (parallelstripsadaptive.go:110-114)
        assert rc == 0 and blob[:4] == b"PICA", name
        assert [int.from_bytes(blob[4 + 4 * k: 8 + 4 * k], "little") for k in range(3)] == [w, h, strips]
        starts = mico.pica_boundaries(img, strips)
        off = 0
        for s in range(strips):
            e = blob[16 + 16 * s: 32 + 16 * s]
            y0, o, ln, fl = (int.from_bytes(e[4 * k: 4 * k + 4], "little") for k in range(4))
            assert (y0, o) == (starts[s], off) and fl in (0, 1)
            y1 = starts[s + 1] if s + 1 < strips else h
            rc, a = mico.compress_single_frame(img[y0:y1], mx, 2)
            rc2, g = mico.compress_single_frame_grad(img[y0:y1], mx)
            assert rc == 0 and rc2 == 0
            want = g if len(g) <= len(a) else a                                # ties go to the gradient predictor (:98)
            assert fl == (1 if len(g) <= len(a) else 0)
            assert blob[16 + 16 * strips + o: 16 + 16 * strips + o + ln] == want
            off += ln
        assert len(blob) == 16 + 16 * strips + off
        rc, back = mico.pica_decompress(blob)
        assert rc == 0 and np.array_equal(back, img), name
        rc, pics = mico.pics_compress(img, mx, strips, 2)
        assert rc == 0
        if name == "MR" and strips > 1:
            assert len(blob) < len(pics)                                       # parallelstripsadaptive_test.go:46-76 reports the gain
    assert mico.pica_decompress(b"PICS" + bytes(32))[0] != 0
    assert mico.pica_decompress(blob[:40])[0] != 0


def test_published_ratios_and_predictor_choices(mico):
    """docs/adaptive-compression.md: PICS-4 / PICA-4 = MR 2.284 / 2.309 with 3 of 4 strips on the gradient predictor, CT 2.145 /
    2.112 with none (:70-83); CompressSingleFrameGrad on CT 2.182 against 2.237 (:33); on MR the gradient predictor gains 1.1 %
    (:32 -- that table's absolute MR figures predate the tableLog-13 rule, the gain does not)."""
    mr = np.fromfile(os.path.join(GOLDEN, "MR_256_256_image.bin"), dtype="<u2").reshape(256, 256)
    ct = np.fromfile(os.path.join(GOLDEN, "CT_512_512_image.bin"), dtype="<u2").reshape(512, 512)
    want = {"MR": (2.284, 2.309, 3), "CT": (2.145, 2.112, 0)}
    for name, img in (("MR", mr), ("CT", ct)):
        mx, raw = int(img.max()), img.size * 2
        rc, pics = mico.pics_compress(img, mx, 4, 2)
        rc2, pica = mico.pica_compress(img, mx, 4)
        assert rc == 0 and rc2 == 0
        grad = sum(int.from_bytes(pica[16 + 16 * s + 12: 32 + 16 * s], "little") for s in range(4))
        assert (round(raw / len(pics), 3), round(raw / len(pica), 3), grad) == want[name], name
    rc, a = mico.compress_single_frame(ct, int(ct.max()), 2)
    rc, g = mico.compress_single_frame_grad(ct, int(ct.max()))
    assert (round(ct.size * 2 / len(a), 3), round(ct.size * 2 / len(g), 3)) == (2.237, 2.182)
    rc, a = mico.compress_single_frame(mr, int(mr.max()), 2)
    rc, g = mico.compress_single_frame_grad(mr, int(mr.max()))
    assert abs(len(a) / len(g) - 1.011) < 0.0006

"""The row-by-row inverse predictor (k_dec_predict_rows, csrc/mic_decode_rows.hip: frames of 1009..2688 columns) against the oracle's
serial decoder (deltarlecompressu16.go:69-128): every chunk class at both ends of its width range, frames of one to a few rows, escapes
in the first and last column and in the lane whose chunk the row's end cuts, and -- the path a stream written by an encoder never
takes -- symbol streams whose pixels wrap around 16 bits, which the kernel must notice and decode with the reference's arithmetic."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (columns) both ends of every chunk class K = 16, 20, 24, 28, 32, 36, 40, 42 pixels per lane
WIDTHS = [1009, 1024, 1025, 1280, 1281, 1536, 1537, 1792, 1793, 2048, 2049, 2304, 2305, 2560, 2561, 2577, 2688]


def _spiky(synth, w, h, depth, seed):
    """XR-like frame with saturated / zero pixels sprinkled in: escapes (raw pixels) in many rows, column 0 and the last column included."""
    img = synth.xr_like(cols=w, rows=h, depth=depth, seed=seed, noise=6.0)
    rng = np.random.default_rng(seed)
    maxv = (1 << depth) - 1
    n = max(4, (w * h) // 300)
    ys, xs = rng.integers(0, h, n), rng.integers(0, w, n)
    img[ys, xs] = np.where(rng.integers(0, 2, n) == 1, maxv, 0).astype(np.uint16)
    img[:, 0] = np.where(np.arange(h) % 3 == 0, maxv, img[:, 0])
    img[:, w - 1] = np.where(np.arange(h) % 4 == 1, 0, img[:, w - 1])
    img[h // 2:, w // 3: w // 3 + 7] = maxv                                 # an edge that runs down the frame
    return img


@pytest.mark.parametrize("w", WIDTHS)
def test_rows_predictor_round_trip_every_chunk_class(mic, mico, synth, gpu_ready, w):
    done = 0
    for h, depth, seed in ((1, 12, 1), (2, 12, 2), (3, 10, 3), (37, 12, 4), (66, 16, 5)):
        for img in (synth.xr_like(cols=w, rows=h, depth=depth, seed=seed), _spiky(synth, w, h, depth, seed + 10)):
            maxv = (1 << depth) - 1
            rc, stream = mico.compress_single_frame(img, maxv, 2)
            if rc:                                                      # (a frame of border rows only: nothing to entropy-code)
                continue
            got = mic.decompress_single_frame(stream, w, h)
            assert np.array_equal(got.reshape(h, w), img), (w, h, depth)
            done += 1
    assert done >= 4


def test_rows_predictor_full_strip_with_escapes(mic, mico, synth, gpu_ready):
    """An XR strip (2577 x 256: K = 42, the last lane's chunk cut at 15 pixels) and a 2688-column one, escapes in every row."""
    for w, h in ((2577, 256), (2688, 130), (1344, 300)):
        img = _spiky(synth, w, h, 12, w)
        rc, stream = mico.compress_single_frame(img, 4095, 2)
        assert rc == 0
        assert mic.compress_single_frame(img, w, h, 4095, 2) == stream
        got = mic.decompress_single_frame(stream, w, h)
        assert np.array_equal(got.reshape(h, w), img)


def test_rows_predictor_wrapping_streams_agree_with_the_oracle(mic, mico, synth, gpu_ready):
    """Symbols no encoder writes (far outside thr +- thr) make pixels wrap around 16 bits -- in the reference that is plain uint16
    arithmetic.  The closed form of the row kernel does not hold there: it must notice and produce the reference's pixels."""
    rng = np.random.default_rng(21)
    checked = differing_from_clean = 0
    for seed, (h, w), depth in ((1, (40, 1100), 12), (2, (20, 2577), 12), (3, (14, 1700), 12), (4, (24, 2688), 12), (5, (90, 1009), 16)):
        img = synth.xr_like(cols=w, rows=h, depth=depth, seed=seed, noise=8.0)
        maxv = (1 << depth) - 1
        tok = mico.delta_rle_compress(img, maxv)
        clean = None
        for k in range(14):
            t = tok.copy()
            if k:
                # literal symbols replaced by arbitrary values below the delimiter: the token structure (headers) stays what it was
                mid = (1 << (int(t[0]).bit_length() - 1)) - 1
                cand = np.nonzero((t[4:] > 8) & (t[4:] < mid))[0] + 4             # values that are neither small counts nor headers
                pick = rng.choice(cand, size=min(cand.size, int(rng.integers(1, 40))), replace=False)
                t[pick] = rng.integers(0, max(int(t[0]) - 1, 2), pick.size).astype(np.uint16)
            rc, stream = mico.fse_compress(t, 2)
            if rc:
                continue
            rc_o, want = mico.decompress_single_frame(stream, w, h)
            try:
                got, rc_g = mic.decompress_single_frame(stream, w, h), 0
            except mic.MicError as e:
                got, rc_g = None, e.code
            assert (rc_g == 0) == (rc_o == 0), (seed, k, rc_g, rc_o)
            if rc_o == 0:
                assert np.array_equal(got, want), (seed, k, int(np.count_nonzero(got != want)))
                if k == 0:
                    clean = want
                elif clean is not None and not np.array_equal(want, clean):
                    differing_from_clean += 1
                checked += 1
    assert checked > 40 and differing_from_clean > 20


def test_wide_frames_with_damaged_tokens_agree_with_the_oracle(mic, mico, synth, gpu_ready):
    """What test_gpu_parity's damaged-token test does, at the widths the fused tokens-to-pixels kernel takes (k_dec_rows_tok: rows
    assembled from the header walk's segments): zero counts, headers that point past the end or into the middle of a chunk, streams
    that stop early -- coded again so that the entropy stage accepts them.  The kernel must decline what it cannot take for certain
    (the two-kernel path then reports the reference's error) and make the oracle's pixels of everything else."""
    rng = np.random.default_rng(17)
    total = decodable = 0
    for seed, noise, (h, w) in ((3, 2.0, (20, 1100)), (4, 60.0, (12, 2577)), (5, 10.0, (16, 1700)), (6, 167.0, (9, 2688)), (7, 4.0, (30, 1009))):
        img = synth.xr_like(cols=w, rows=h, depth=12, seed=seed, noise=noise)
        tok = mico.delta_rle_compress(img, 4095)
        for k in range(48):
            t = tok.copy()
            for _ in range(int(rng.integers(1, 5))):
                i = int(rng.integers(1, t.size))
                t[i] = (0, 1, 2, int(t[0]), int(t[0]) // 2 + 1, int(rng.integers(0, int(t[0]) + 1)))[int(rng.integers(0, 6))]
            if k % 8 == 2:
                t = t[: int(rng.integers(3, t.size))]
            rc, stream = mico.fse_compress(t, 2)
            if rc:
                continue
            rc_o, want = mico.decompress_single_frame(stream, w, h)
            try:
                got, rc_g = mic.decompress_single_frame(stream, w, h), 0
            except mic.MicError as e:
                got, rc_g = None, e.code
            assert (rc_g == 0) == (rc_o == 0), (seed, k, rc_g, rc_o)
            if rc_o == 0:
                assert np.array_equal(got, want), (seed, k, int(np.count_nonzero(got != want)))
                decodable += 1
            total += 1
    assert total > 100 and decodable > 30 and total - decodable > 20

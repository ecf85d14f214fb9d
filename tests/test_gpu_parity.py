"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI of
libmic_hip.so and is compared bit-for-bit with the CPU oracle and the golden vectors."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(GOLDEN, "golden.json")))


def _mr():
    return np.fromfile(os.path.join(GOLDEN, "MR_256_256_image.bin"), dtype="<u2").reshape(256, 256)


def _ct():
    return np.fromfile(os.path.join(GOLDEN, "CT_512_512_image.bin"), dtype="<u2").reshape(512, 512)


@pytest.mark.parametrize("name", ["MR", "CT"])
@pytest.mark.parametrize("ns", [2, 4, 8])
def test_frame_matches_reference_c_golden(mic, mico, gpu_ready, name, ns):
    """Bit-identical stream vs the reference C codec (tests/golden/golden.json) on the
    reference's own test images; lossless round trip on the GPU."""
    img = _mr() if name == "MR" else _ct()
    rec = GOLD["streams"][f"{name}/{ns}"]
    blob = mic.compress_single_frame(img, rec["width"], rec["height"], rec["max_value"], ns)
    assert len(blob) == rec["size"]
    assert f"{mico.fnv1a64(blob):016x}" == rec["fnv1a64"]
    if name == "MR":
        assert blob == open(os.path.join(GOLDEN, f"MR_256_256_{ns}state.mic"), "rb").read()
    px = mic.decompress_single_frame(blob, rec["width"], rec["height"])
    assert np.array_equal(px, img)


@pytest.mark.parametrize("depth,w,h", [(12, 257, 63), (12, 640, 200), (16, 300, 211), (10, 129, 130), (8, 256, 256)])
@pytest.mark.parametrize("ns", [2, 4, 8])
def test_frame_matches_oracle_synthetic(mic, mico, synth, gpu_ready, depth, w, h, ns):
    img = synth.xr_like(cols=w, rows=h, depth=depth, seed=depth * 7 + ns)
    maxv = (1 << depth) - 1                      # caller-supplied maxValue > image max is legal
    rc, want = mico.compress_single_frame(img, maxv, ns)
    assert rc == 0
    got = mic.compress_single_frame(img, w, h, maxv, ns)
    assert got == want
    assert np.array_equal(mic.decompress_single_frame(got, w, h), img)


def test_decodes_every_flavour_the_oracle_writes(mic, mico, synth, gpu_ready):
    """FSEDecompressU16Auto: 1-state (no prefix), FF02, FF04, FF84 and rANS FF08."""
    img = synth.xr_like(cols=200, rows=150, depth=12, seed=5)
    tok = mico.delta_rle_compress(img, 4095)
    for ns in (1, 2, 4, 8, 108):
        rc, blob = mico.fse_compress(tok, ns)
        assert rc == 0
        px = mic.decompress_single_frame(blob, 200, 150)
        assert np.array_equal(px, img), ns


def test_constant_frame_is_use_rle(mic, mico, gpu_ready):
    """All tokens equal is impossible for a frame (delimiter + maxValue differ), but a constant
    frame still exercises the long same-run path."""
    img = np.full((64, 500), 1234, dtype=np.uint16)
    rc, want = mico.compress_single_frame(img, 4095, 2)
    if rc == 0:
        assert mic.compress_single_frame(img, 500, 64, 4095, 2) == want
    else:
        with pytest.raises(mic.MicError) as e:
            mic.compress_single_frame(img, 500, 64, 4095, 2)
        assert e.value.code == rc


def test_noise_frame_error_matches_oracle(mic, mico, synth, gpu_ready):
    noise = (synth.hash_u64(512 * 512, 3) & np.uint64(0xFFFF)).astype(np.uint16).reshape(512, 512)
    rc, want = mico.compress_single_frame(noise, 65535, 2)
    if rc == 0:
        assert mic.compress_single_frame(noise, 512, 512, 65535, 2) == want
    else:
        with pytest.raises(mic.MicError) as e:
            mic.compress_single_frame(noise, 512, 512, 65535, 2)
        assert e.value.code == rc


def test_escape_path_and_long_runs(mic, mico, gpu_ready):
    """|diff| >= T escapes (deltarlecompressu16.go:52-56) and run chunking at midCount
    (rlecompressu16.go:57-67, midCount = 127 for depth 8)."""
    img = np.zeros((96, 700), dtype=np.uint16)
    img[:, ::3] = 255
    img[40:60, :] = 7
    img[60:, 100:600] = np.arange(500, dtype=np.uint16)[None, :] % 256
    for maxv in (255, 511, 4095):
        rc, want = mico.compress_single_frame(img, maxv, 2)
        assert rc == 0
        got = mic.compress_single_frame(img, 700, 96, maxv, 2)
        assert got == want
        assert np.array_equal(mic.decompress_single_frame(got, 700, 96), img)


@pytest.mark.parametrize("strips", [1, 2, 4, 8])
def test_pics_matches_oracle(mic, mico, gpu_ready, strips):
    """CompressParallelStrips: global maxValue for every strip (parallelstrips.go:88)."""
    img = _mr()
    maxv = int(img.max())
    rc, want = mico.pics_compress(img, maxv, strips, 2)
    assert rc == 0
    got = mic.compress_parallel_strips(img, 256, 256, maxv, strips)
    assert got == want
    px, w, h = mic.decompress_parallel_strips(got)
    assert (w, h) == (256, 256) and np.array_equal(px, img)


@pytest.mark.parametrize("ns", [4, 8])
def test_pics_nstate_variants(mic, mico, gpu_ready, ns):
    img = _mr()
    rc, want = mico.pics_compress(img, int(img.max()), 8, ns)
    assert rc == 0
    assert mic.compress_parallel_strips(img, 256, 256, int(img.max()), 8, ns) == want
    px, _, _ = mic.decompress_parallel_strips(want)
    assert np.array_equal(px, img)


def test_pics_clamp_bad_magic_truncation(mic, gpu_ready):
    full = _mr()
    img = full[:2].copy()
    blob = mic.compress_parallel_strips(img, 256, 2, int(full.max()), 256)   # parallelstrips_test.go:119-145
    assert mic.pics_info(blob)[2] == 2
    px, w, h = mic.decompress_parallel_strips(blob)
    assert (w, h) == (256, 2) and np.array_equal(px, img)
    with pytest.raises(mic.MicError):
        mic.decompress_parallel_strips(b"XXXX" + blob[4:])                   # :91-96
    with pytest.raises(mic.MicError):
        mic.decompress_parallel_strips(blob[:10])                            # :97-102


def test_pics_rows_no_strip_covers_come_back_zero(mic, mico, gpu_ready):
    """A PICS header whose strips stop short of the image height is accepted by the reference, and the rows nobody writes
    are the zeros of make([]uint16, w*h) (parallelstrips.go:288-320) -- not whatever an earlier call left in the staging buffer."""
    img = _mr()[:64].copy()
    mic.decompress_parallel_strips(mic.compress_parallel_strips(_mr(), 256, 256, int(_mr().max()), 8))   # leave pixels behind
    blob = bytearray(mic.compress_parallel_strips(img, 256, 64, int(img.max()), 4))
    blob[8:12] = (80).to_bytes(4, "little")                                  # height 80, 4 strips of 16 rows: rows 64-79 uncovered
    px, w, h = mic.decompress_parallel_strips(bytes(blob))
    assert (w, h) == (256, 80) and np.array_equal(px[:64], img) and not px[64:].any()
    rc, want = mico.pics_decompress(bytes(blob))
    assert rc == 0 and want.shape == (80, 256) and np.array_equal(want, px)


def test_mic3_level_table_is_validated(mic, synth, gpu_ready):
    """Level descriptors that are not what computeLevels writes (wsiformat.go:244-271) are refused instead of steering the tile loops."""
    blob = bytearray(mic.compress_wsi(synth.wsi_like(300, 200, seed=3), 300, 200))
    for off, val in ((48 + 8, 1 << 30), (48 + 12, 0), (48 + 16, 5000), (48 + 0, 0)):   # tilesX, tilesY, firstTileIdx, width of level 0
        bad = bytearray(blob)
        bad[off:off + 4] = int(val).to_bytes(4, "little")
        for call in (lambda b: mic.decompress_wsi_level(b, 0), lambda b: mic.decompress_wsi_tile(b, 0, 0, 0),
                     lambda b: mic.decompress_wsi_region(b, 0, 0, 0, 10, 10)):
            with pytest.raises(mic.MicError) as e:
                call(bytes(bad))
            assert e.value.code == mic.MIC_ERR_CORRUPT


def test_fse_scratch_knobs(mic, mico, gpu_ready):
    """ScratchU16.TableLog and .DecompressLimit on the bare FSE calls (fseu16.go:87-102): the table log the caller asks for is where
    optimalTableLog starts (fsecompressu16.go:480-518), and the limit is compared at every wrap of the 65536-symbol ring."""
    tok = mico.delta_rle_compress(_mr(), int(_mr().max()))                   # 65 578 tokens
    for ns in (1, 2, 4):
        for tl in (0, 9, 12, 14, 16):
            rc, want = mico.fse_compress_tl(tok, ns, tl)
            assert rc == 0 and mic.fse_compress_u16(tok, ns, table_log=tl) == want
            assert np.array_equal(mic.fse_decompress_u16_auto(want, len(tok) + 8), tok)
    assert mic.fse_compress_u16(tok, 2, table_log=14)[6] & 15 == 14 - 5 and mic.fse_compress_u16(tok, 2)[6] & 15 == 13 - 5
    with pytest.raises(mic.MicError) as e:
        mic.fse_compress_u16(tok, 2, table_log=17)                           # "tableLog (17) > maxTableLog (16)"
    assert e.value.code == mic.MIC_ERR_ARGS
    blob = mic.fse_compress_u16(tok, 2)
    assert np.array_equal(mic.fse_decompress_u16_auto(blob, len(tok) + 8, decompress_limit=65537), tok)   # one wrap at 65536 < limit
    with pytest.raises(mic.MicError) as e:
        mic.fse_decompress_u16_auto(blob, len(tok) + 8, decompress_limit=65536)
    assert e.value.code == mic.MIC_ERR_CAPACITY
    short = mic.fse_compress_u16(tok[:30000], 2)
    assert len(mic.fse_decompress_u16_auto(short, 30008, decompress_limit=100)) == 30000        # no wrap, no check (N-state)
    one = mic.fse_compress_u16(tok[:30000], 1)
    with pytest.raises(mic.MicError):
        mic.fse_decompress_u16_auto(one, 30008, decompress_limit=30000)      # 1-state: checked at the end, fsedecompressu16.go:372
    assert len(mic.fse_decompress_u16_auto(one, 30008, decompress_limit=30001)) == 30000


def test_mic2_matches_oracle(mic, mico, synth, gpu_ready):
    stack = synth.ct_stack(frames=6, size=128, depth=12, seed=21)
    rc, want = mico.mic2_compress(stack, 4095, False)
    assert rc == 0
    got = mic.compress_multi_frame(stack, 128, 128, 4095)
    assert got == want
    assert np.array_equal(mic.decompress_multi_frame(got), stack)


def test_mic2_temporal_matches_oracle(mic, mico, synth, gpu_ready):
    """Temporal pipeline (multiframecompress.go:179-261): byte-identical container, both decoders agree, and the
    GPU decodes what the oracle wrote."""
    stack = synth.ct_stack(frames=7, size=128, depth=12, seed=33)
    rc, want = mico.mic2_compress(stack, 4095, True)
    assert rc == 0 and want[16] == 0x03
    got = mic.compress_multi_frame(stack, 128, 128, 4095, temporal=True)
    assert got == want
    assert np.array_equal(mic.decompress_multi_frame(want), stack)
    rc, back = mico.mic2_decompress(got)
    assert rc == 0 and np.array_equal(np.asarray(back).reshape(stack.shape), stack)
    # wrap-around residuals: a frame pair that differs by more than 32767
    f0 = synth.xr_like(cols=160, rows=96, depth=12, seed=5).astype(np.int64)
    n1 = (synth.hash_u64(96 * 160, 77) % np.uint64(41)).astype(np.int64).reshape(96, 160)
    n2 = (synth.hash_u64(96 * 160, 78) % np.uint64(29)).astype(np.int64).reshape(96, 160)
    f1 = (f0 + 40000 + n1) % 65536
    big = np.stack([f0, f1, (f1 - 39000 + n2) % 65536]).astype(np.uint16)
    rc, want = mico.mic2_compress(big, 65535, True)
    assert rc == 0
    got = mic.compress_multi_frame(big, 160, 96, 65535, temporal=True)
    assert got == want
    assert np.array_equal(mic.decompress_multi_frame(got), big)
    # truncated residual stream
    with pytest.raises(mic.MicError):
        mic.decompress_multi_frame(want[:-3])
    # DecompressFrame: any frame of either pipeline (multiframecompress.go:266-315)
    rc, indep = mico.mic2_compress(stack, 4095, False)
    rc, temp = mico.mic2_compress(stack, 4095, True)
    for idx in (0, 3, 6):
        assert np.array_equal(mic.decompress_frame(indep, idx), stack[idx])
        assert np.array_equal(mic.decompress_frame(temp, idx), stack[idx])
    with pytest.raises(mic.MicError):
        mic.decompress_frame(temp, 7)


def test_batch_mixed_shapes(mic, mico, synth, gpu_ready):
    """One batch, ragged shapes; tiny noisy frames fail in the oracle (normaliser error) and
    must fail with the same code on the GPU."""
    frames = [synth.xr_like(cols=100 + 97 * i, rows=40 + 61 * i, depth=12, seed=30 + i) for i in range(7)]
    res = mic.compress_batch(frames, [4095] * 7, 2)
    ok = []
    for f, (st, blob, used) in zip(frames, res):
        rc, want = mico.compress_single_frame(f, 4095, 2)
        assert st == rc and blob == want
        if rc == 0:
            ok.append((f, blob))
    assert len(ok) >= 4
    outs = mic.decompress_batch([b for _, b in ok], [(f.shape[1], f.shape[0]) for f, _ in ok])
    for (f, _), (st, px) in zip(ok, outs):
        assert st == 0 and np.array_equal(px, f)


def test_decode_batch_of_mixed_flavours(mic, mico, synth, gpu_ready):
    """One decode batch whose neighbouring units differ in every way the kernels specialise on: 2/4/8-state and rANS-free
    1-state fallbacks, tableLog 13 next to tableLog 16, very different lengths, an odd unit count (the two-streams-per-wave
    decoder pairs units 2w / 2w+1 and must leave a foreign or missing partner alone)."""
    frames, blobs = [], []
    specs = [(12, 300, 200, 2), (12, 260, 90, 4), (16, 256, 256, 2), (12, 500, 30, 8), (8, 256, 256, 2),
             (12, 31, 9, 2), (16, 200, 180, 4), (10, 640, 64, 8), (12, 2577, 64, 2)]
    for i, (depth, w, h, ns) in enumerate(specs):
        img = synth.xr_like(cols=w, rows=h, depth=depth, seed=200 + i)
        rc, blob = mico.compress_single_frame(img, (1 << depth) - 1, ns)
        if rc != 0:
            continue
        frames.append(img); blobs.append(blob)
    assert len(frames) >= 7
    outs = mic.decompress_batch(blobs, [(f.shape[1], f.shape[0]) for f in frames])
    for f, (st, px) in zip(frames, outs):
        assert st == 0 and np.array_equal(px, f)
    # the same units in reverse order (pairs differently)
    outs = mic.decompress_batch(blobs[::-1], [(f.shape[1], f.shape[0]) for f in frames[::-1]])
    for f, (st, px) in zip(frames[::-1], outs):
        assert st == 0 and np.array_equal(px, f)
    # drop the first unit: every pairing shifts by one
    outs = mic.decompress_batch(blobs[1:], [(f.shape[1], f.shape[0]) for f in frames[1:]])
    for f, (st, px) in zip(frames[1:], outs):
        assert st == 0 and np.array_equal(px, f)


def test_corrupt_stream_is_an_error_not_a_hang(mic, mico, gpu_ready):
    img = _mr()
    rc, blob = mico.compress_single_frame(img, int(img.max()), 2)
    bad = bytearray(blob)
    bad[-1] = 0                                   # no end mark (bitreader.go:36-38)
    with pytest.raises(mic.MicError):
        mic.decompress_single_frame(bytes(bad), 256, 256)
    with pytest.raises(mic.MicError):
        mic.decompress_single_frame(blob[:40], 256, 256)


def _piecewise_rows(synth, w, h, seed, alphabet, maxseg):
    """Piecewise-constant rows with random segment lengths: dense in runs of every length around
    the tokeniser's chunk size, plus isolated symbols between runs."""
    r = synth.hash_u64(w * h * 2, seed)
    vals = (r[: w * h] % np.uint64(alphabet)).astype(np.uint16)
    lens = (r[w * h:] % np.uint64(maxseg)).astype(np.int64) + 1
    out = np.empty(w * h, dtype=np.uint16)
    pos = 0; k = 0
    while pos < w * h:
        n = int(lens[k]); out[pos:pos + n] = vals[k]; pos += n; k += 1
    return out[: w * h].reshape(h, w)


@pytest.mark.parametrize("maxv", [15, 31, 63, 127, 255, 1023])
def test_tokeniser_fuzz_small_depths(mic, mico, synth, gpu_ready, maxv):
    """Small depths make midCount tiny (7, 15, 31, 63, 127, 511): same-run and literal chunking
    (rlecompressu16.go:57-67) fire constantly, including at the end of the stream.  63 and 127 (chunks of 28 and 60) are the shortest
    the closed-form count / position-per-lane write of the tokeniser takes: a count wraps in most threads there."""
    frames, want = [], []
    for k in range(24):
        w, h = 97 + 13 * k, 40 + (k % 5) * 9
        f = _piecewise_rows(synth, w, h, 1000 + 31 * k + maxv, min(maxv, 6 + k % 7), 1 + (k * 7) % 40)
        if k % 3 == 0:
            f[:, : w // 3] = (np.arange(w // 3) * 7 % (maxv + 1)).astype(np.uint16)[None, :]   # long literal stretches
        if k % 4 == 1:
            f[-1, -(k % 6 + 1):] = (f[-1, -(k % 6 + 1):] + 1 + np.arange(k % 6 + 1)) % (maxv + 1)   # ragged stream end
        frames.append(f)
        want.append(mico.compress_single_frame(f, maxv, 2))
    got = mic.compress_batch(frames, [maxv] * len(frames), 2)
    n_ok = 0
    for f, (st, blob, used), (rc, w_blob) in zip(frames, got, want):
        assert st == rc, (st, rc, f.shape)
        assert blob == w_blob, f.shape
        if rc == 0:
            n_ok += 1
            assert np.array_equal(mic.decompress_single_frame(blob, f.shape[1], f.shape[0]), f)
    assert n_ok >= 12


@pytest.mark.parametrize("ns", [2, 4, 8])
def test_tiny_and_ragged_frames(mic, mico, synth, gpu_ready, ns):
    """Token counts around the lane count and segment boundaries of the parallel encoder; whatever
    the oracle says (blob or error code, incl. the N -> ... -> 1 fallback) the GPU must say too."""
    frames = []
    for k, (w, h) in enumerate([(1, 1), (2, 1), (3, 1), (1, 7), (5, 2), (9, 1), (17, 3), (33, 2), (64, 1), (127, 3),
                                (256, 4), (511, 2), (1023, 1), (1025, 3), (2049, 1), (4097, 2)]):
        f = _piecewise_rows(synth, w, h, 77 + k, 5, 9)
        frames.append(f)
    got = mic.compress_batch(frames, [255] * len(frames), ns)
    for f, (st, blob, used) in zip(frames, got):
        rc, want = mico.compress_single_frame(f, 255, ns)
        assert st == rc, (f.shape, st, rc)
        assert blob == want, f.shape
        if rc == 0:
            assert np.array_equal(mic.decompress_single_frame(blob, f.shape[1], f.shape[0]), f)


def test_sixteen_bit_depth_full_alphabet(mic, mico, synth, gpu_ready):
    """maxValue 65535 -> delimiter 65535, 65536-symbol alphabet, tableLog 16 (the CT case)."""
    img = synth.xr_like(cols=700, rows=300, depth=16, seed=9)
    for ns in (2, 4, 8):
        rc, want = mico.compress_single_frame(img, 65535, ns)
        assert rc == 0
        got = mic.compress_single_frame(img, 700, 300, 65535, ns)
        assert got == want
        assert np.array_equal(mic.decompress_single_frame(got, 700, 300), img)


def test_strip_sized_unit_matches_oracle(mic, mico, synth, gpu_ready):
    """One full XR strip (2577 x 256): the shape bench.py runs, bit-exact against the oracle."""
    img = synth.xr_like(cols=2577, rows=256, depth=12, seed=3)
    rc, want = mico.compress_single_frame(img, 4095, 2)
    assert rc == 0
    got = mic.compress_single_frame(img, 2577, 256, 4095, 2)
    assert got == want
    assert np.array_equal(mic.decompress_single_frame(got, 2577, 256), img)


# ---- bare FSE stage: fse2state_test.go, fse4state_test.go, fse8state_test.go, rans8state_test.go ----
@pytest.mark.parametrize("flavour", [1, 2, 4, 8, 108])
def test_fse_stage_matches_oracle(mic, mico, synth, gpu_ready, flavour):
    img = synth.xr_like(cols=500, rows=180, depth=12, seed=17)
    tok = mico.delta_rle_compress(img, 4095)
    rc, want = mico.fse_compress(tok, flavour)
    assert rc == 0
    got = mic.fse_compress_u16(tok, flavour)
    assert got == want
    if flavour != 1:
        assert got[:2] == bytes([0xFF, {2: 0x02, 4: 0x04, 8: 0x84, 108: 0x08}[flavour]])   # fse2state_test.go:117-142
    back = mic.fse_decompress_u16_auto(got, tok.size + 16)
    assert np.array_equal(back, tok)


@pytest.mark.parametrize("flavour", [1, 2, 4, 8, 108])
def test_fse_stage_edge_cases(mic, mico, gpu_ready, flavour):
    """all-same -> ErrUseRLE; two elements -> error; lengths n%4, n%8 != 0 with i%8 / i%17 data
    (fse2state_test.go:178-257, fse4state_test.go:105-, fse8state_test.go:106-, rans8state_test.go:105-147)."""
    with pytest.raises(mic.ErrUseRLE):
        mic.fse_compress_u16(np.full(1000, 42, dtype=np.uint16), flavour)
    with pytest.raises(mic.MicError):
        mic.fse_compress_u16(np.array([1, 2], dtype=np.uint16), flavour)
    for n in list(range(9, 16)) + list(range(101, 108)) + list(range(1001, 1008)) + [4096, 70001]:
        for mod in (8, 17):
            data = (np.arange(n) % mod).astype(np.uint16)
            rc, want = mico.fse_compress(data, flavour)
            if rc != 0:
                with pytest.raises(mic.MicError) as e:
                    mic.fse_compress_u16(data, flavour)
                assert e.value.code == rc, (n, mod)
                continue
            got = mic.fse_compress_u16(data, flavour)
            assert got == want, (n, mod)
            assert np.array_equal(mic.fse_decompress_u16_auto(got, n + 8), data)


@pytest.mark.parametrize("flavour", [2, 8])
def test_fse_stage_sparse_alphabets_ncount_run_codes(mic, mico, synth, gpu_ready, flavour):
    """Zero-runs of 1, 2, 3, 23, 24, 25, 71, 72 and ~3000 symbols in the normalised counts: the NCount header's
    0xFFFF / '11' / 2-bit run codes (fsecompressu16.go:207-236) at every boundary."""
    r = synth.hash_u64(60000, 5)
    for gaps in ([1, 2, 3, 4], [23, 24, 25, 26], [71, 72, 73, 3000], [48, 96, 24 * 5 + 3, 24 * 7 + 2]):
        vals = np.cumsum([3] + [g + 1 for g in gaps]).astype(np.uint16)        # gaps[i] zeros between used symbols
        vals = np.concatenate([[0, 1, 2], vals]).astype(np.uint16)
        # skewed use so that counts differ (some symbols become -1)
        idx = np.minimum((r % np.uint64(97)) % np.uint64(len(vals) * 3), np.uint64(len(vals) - 1)).astype(np.int64)
        data = vals[idx]
        rc, want = mico.fse_compress(data, flavour)
        assert rc == 0, gaps
        got = mic.fse_compress_u16(data, flavour)
        assert got == want, gaps
        assert np.array_equal(mic.fse_decompress_u16_auto(got, data.size + 8), data)


@pytest.mark.parametrize("w,h", [(8200, 3), (9001, 70), (33000, 2), (40000, 3)])
def test_wide_frames_row_buffer_classes(mic, mico, synth, gpu_ready, w, h):
    """Frames wider than the ordinary row-buffer class of the predictor kernel, and wider than any (fallback path)."""
    noise = (synth.hash_u64(w * h, w + h) % np.uint64(5)).astype(np.int64).reshape(h, w)
    img = ((np.add.outer(np.arange(h) * 7, np.arange(w) // 8) + noise) % 4096).astype(np.uint16)
    rc, want = mico.compress_single_frame(img, 4095, 2)
    assert rc == 0
    got = mic.compress_single_frame(img, w, h, 4095, 2)
    assert got == want
    assert np.array_equal(np.asarray(mic.decompress_single_frame(got, w, h)).reshape(-1), img.reshape(-1))


def test_fse_stage_incompressible_noise(mic, synth, gpu_ready):
    noise = (synth.hash_u64(1 << 21, 9) & np.uint64(0xFFFF)).astype(np.uint16)
    for flavour in (2, 108):
        with pytest.raises(mic.ErrIncompressible):
            mic.fse_compress_u16(noise, flavour)


# ---- MIC3 / WSI: wsi_test.go:361-603 (tile kinds, full slide, odd dimensions) ------------------------
def test_wsi_matches_oracle_and_round_trips(mic, mico, synth, gpu_ready):
    img = synth.wsi_like(600, 420, seed=5)                   # 3 levels, edge tiles, white + tissue tiles
    rc, want = mico.wsi_compress(img)
    assert rc == 0
    got = mic.compress_wsi(img, 600, 420)
    assert got == want
    hdr = mic.read_wsi_header(got)
    assert [(l["width"], l["height"]) for l in hdr["levels"]] == [(600, 420), (300, 210), (150, 105)]
    assert np.array_equal(mic.decompress_wsi_level(got, 0), img)
    for lvl in range(3):
        for ty in range(hdr["levels"][lvl]["tiles_y"]):
            for tx in range(hdr["levels"][lvl]["tiles_x"]):
                rc, t = mico.wsi_decompress_tile_at(got, lvl, tx, ty)
                assert rc == 0
                assert np.array_equal(mic.decompress_wsi_tile(got, lvl, tx, ty), t)


def test_wsi_region_matches_the_slide(mic, synth, gpu_ready):
    """DecompressWSIRegion (wsicompress.go:219-297): rectangles inside a tile, across tile borders, clamped at the edge,
    on level 0 (= the original pixels: the codec is lossless) and on level 1 (= the same cut of the whole-level decode)."""
    W, H = 700, 520
    slide = synth.wsi_like(W, H, seed=9)
    blob = mic.compress_wsi(slide, W, H, tile_w=256, tile_h=256, levels=2)
    for (x, y, w, h) in [(10, 20, 100, 60), (200, 200, 200, 200), (0, 0, W, H), (600, 400, 300, 300), (255, 255, 2, 2)]:
        got = mic.decompress_wsi_region(blob, 0, x, y, w, h)
        assert np.array_equal(got, slide[y:y + h, x:x + w])
    lv1 = mic.decompress_wsi_level(blob, 1).reshape(H // 2, W // 2, 3)
    got = mic.decompress_wsi_region(blob, 1, 100, 50, 200, 150)
    assert np.array_equal(got, lv1[50:200, 100:300])
    with pytest.raises(mic.MicError):
        mic.decompress_wsi_region(blob, 0, W, 0, 10, 10)            # empty after clamping
    with pytest.raises(mic.MicError):
        mic.decompress_wsi_region(blob, 5, 0, 0, 10, 10)


@pytest.mark.parametrize("bits", [8, 16])
def test_wsi_greyscale_matches_oracle(mic, mico, synth, gpu_ready, bits):
    """CompressWSI with channels = 1 (wsicompress.go:58-69, :366-370): Downsample2xGrey pyramid, one plane per tile, bare plane blobs;
    tile / level / region decode (decompressGreyTileBlob, :477-484) in the slide's sample width."""
    from test_oracle_wavelet_wsi import _grey_slide
    W, H = 600, 420
    img = _grey_slide(synth, W, H, bits, seed=6)
    rc, want = mico.wsi_compress_grey(img)
    assert rc == 0
    got = mic.compress_wsi(img, W, H, channels=1, bits_per_sample=bits)
    assert got == want
    hdr = mic.read_wsi_header(got)
    assert (hdr["channels"], hdr["bits_per_sample"], hdr["color_transform"]) == (1, bits, False) and len(hdr["levels"]) == 3
    lv0 = mic.decompress_wsi_level(got, 0)
    assert lv0.dtype == img.dtype and np.array_equal(lv0, img)
    for lvl, L in enumerate(hdr["levels"]):
        for ty in range(L["tiles_y"]):
            for tx in range(L["tiles_x"]):
                rc, t = mico.wsi_decompress_tile_at(got, lvl, tx, ty)
                assert rc == 0 and np.array_equal(mic.decompress_wsi_tile(got, lvl, tx, ty), t)
    a = img.astype(np.uint32)
    lv1 = ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) // 4).astype(img.dtype)
    assert np.array_equal(mic.decompress_wsi_level(got, 1), lv1)
    assert np.array_equal(mic.decompress_wsi_region(got, 0, 200, 100, 300, 250), img[100:350, 200:500])
    assert np.array_equal(mic.decompress_wsi_region(got, 1, 250, 200, 500, 500), lv1[200:, 250:])
    # odd tile shape, explicit level count
    rc, want = mico.wsi_compress_grey(img, 200, 100, 2)
    assert rc == 0 and mic.compress_wsi(img, W, H, channels=1, bits_per_sample=bits, tile_w=200, tile_h=100, levels=2) == want


def test_wsi_greyscale_noise_errors_and_raw_planes(mic, mico, synth, gpu_ready):
    from test_oracle_wavelet_wsi import grey_raw_container
    n16 = (synth.hash_u64(256 * 256, 31).reshape(256, 256) & np.uint64(0xFFFF)).astype(np.uint16)
    n14 = n16 & np.uint16(0x3FFF)
    rc, want = mico.wsi_compress_grey(n14)
    assert rc == 0 and mic.compress_wsi(n14, 256, 256, channels=1, bits_per_sample=16) == want
    assert np.array_equal(mic.decompress_wsi_level(want, 0), n14)
    rc, _ = mico.wsi_compress_grey(n16)                        # the reference's normaliser gives up on 16-bit noise
    with pytest.raises(mic.MicError) as e:
        mic.compress_wsi(n16, 256, 256, channels=1, bits_per_sample=16)
    assert rc != 0 and e.value.code == rc
    for img in (n16[:40, :56], (n16[:40, :56] >> 8).astype(np.uint8)):    # planeRaw in a grey tile, wsicompress.go:515-523
        blob = grey_raw_container(np.ascontiguousarray(img))
        assert np.array_equal(mic.decompress_wsi_tile(blob, 0, 0, 0), img)
        assert np.array_equal(mic.decompress_wsi_region(blob, 0, 5, 7, 20, 11), img[7:18, 5:25])
    with pytest.raises(mic.MicError) as e:                     # RGB needs 8 bits (compressTileBlob would misread the bytes)
        mic.compress_wsi(np.zeros((8, 8, 3), np.uint16), 8, 8, channels=3, bits_per_sample=16)
    assert e.value.code == mic.MIC_ERR_UNSUPPORTED


def test_rgb_and_single_frame_files(mic, mico, synth, gpu_ready):
    """CompressRGB / DecompressRGB (rgbcompress.go:25-33) and the CLI's MICR / MIC1 files (cmd/mic-compress/main.go:26-91)."""
    img = np.ascontiguousarray(synth.wsi_like(300, 200, seed=3))
    rc, want = mico.wsi_compress_tile(img)
    assert rc == 0
    got = mic.compress_rgb(img, 300, 200)
    assert got == want and np.array_equal(mic.decompress_rgb(got, 300, 200), img)
    rc, wantf = mico.micr_write(img)
    assert rc == 0 and mic.compress_rgb(img, 300, 200, container=True) == wantf
    assert np.array_equal(mic.decompress_rgb(wantf), img)
    white = np.full((64, 48, 3), 255, np.uint8)                                  # three constant planes: 12 + 3 + 1 + 1 bytes
    assert len(mic.compress_rgb(white, 48, 64)) == 17 and np.array_equal(mic.decompress_rgb(mic.compress_rgb(white, 48, 64), 48, 64), white)
    big = np.ascontiguousarray(synth.wsi_like(1920, 1080, seed=12))              # one 2-megapixel "tile" per plane
    rc, want = mico.wsi_compress_tile(big)
    assert rc == 0 and mic.compress_rgb(big, 1920, 1080) == want
    assert np.array_equal(mic.decompress_rgb(want, 1920, 1080), big)
    mr = np.fromfile(os.path.join(GOLDEN, "MR_256_256_image.bin"), dtype="<u2").reshape(256, 256)
    for ns in (2, 4, 8):
        rc, wantf = mico.mic1_write(mr, int(mr.max()), ns)
        assert rc == 0 and mic.write_mic1(mr, 256, 256, int(mr.max()), ns) == wantf
        assert np.array_equal(mic.read_mic1(wantf), mr)
    for bad in (b"MIC1" + bytes(16), b"MIC1" + bytes(8) + (1).to_bytes(4, "little") + (99).to_bytes(4, "little"), b"MICR" + bytes(8), b"nope"):
        with pytest.raises(mic.MicError):
            (mic.read_mic1 if bad[:4] == b"MIC1" else mic.decompress_rgb)(bad)


def test_wsi_white_slide_is_101_bytes(mic, gpu_ready):
    white = np.full((256, 256, 3), 255, dtype=np.uint8)
    blob = mic.compress_wsi(white, 256, 256)
    assert len(blob) == 101                                  # docs/compression-results.md:167-174 (1946x)
    assert np.array_equal(mic.decompress_wsi_level(blob, 0), white)


def test_wsi_noise_tile_raw_fallback_and_small_tiles(mic, mico, synth, gpu_ready):
    """Noise planes: ErrIncompressible -> raw plane fallback (wsicompress.go:403-414).  With tiny noisy
    tiles the reference's normaliser fails outright (oracle: internal error); the GPU must agree."""
    h = synth.hash_u64(300 * 200 * 3, 77)
    noise = (h & np.uint64(0xFF)).astype(np.uint8).reshape(200, 300, 3)
    for tw, th in ((0, 0), (128, 64)):
        rc, want = mico.wsi_compress(noise, tw or 256, th or 256)
        if rc != 0:
            with pytest.raises(mic.MicError) as e:
                mic.compress_wsi(noise, 300, 200, tile_w=tw, tile_h=th)
            assert e.value.code == rc
            continue
        got = mic.compress_wsi(noise, 300, 200, tile_w=tw, tile_h=th)
        assert got == want
        assert np.array_equal(mic.decompress_wsi_level(got, 0), noise)
    # smooth content with small tiles exercises many tiles per level and cropped edges
    img = synth.wsi_like(333, 217, seed=8)
    rc, want = mico.wsi_compress(img, 200, 100)
    assert rc == 0
    got = mic.compress_wsi(img, 333, 217, tile_w=200, tile_h=100)
    assert got == want
    assert np.array_equal(mic.decompress_wsi_level(got, 0), img)
    assert np.array_equal(mic.decompress_wsi_level(got, 1), mic.decompress_wsi_level(want, 1))


# ---- WaveletV2: waveletu16_test.go:190-248, :385-417; results/.../06-wavelet-simd.txt ---------------------
@pytest.mark.parametrize("name,ratio", [("MR", 2.381), ("CT", 1.669)])
def test_wavelet_v2_matches_oracle_on_reference_images(mic, mico, gpu_ready, name, ratio):
    img = _mr() if name == "MR" else _ct()
    rows, cols = img.shape
    rc, want = mico.wavelet_v2_compress(img, int(img.max()), 5)
    assert rc == 0
    got = mic.wavelet_v2_compress(img, rows, cols, int(img.max()), 5)
    assert got == want
    assert abs(img.size * 2 / len(got) - ratio) < 0.0015        # published WaveletV2 (5 levels) ratio
    px, r, c = mic.wavelet_v2_decompress(got)
    assert (r, c) == (rows, cols) and np.array_equal(px, img)


@pytest.mark.parametrize("rows,cols,levels", [(63, 65, 5), (130, 70, 8), (214, 176, 5), (256, 256, 1), (301, 97, 3)])
def test_wavelet_v2_odd_dims_and_levels(mic, mico, synth, gpu_ready, rows, cols, levels):
    img = synth.xr_like(cols=cols, rows=rows, depth=12, seed=rows * 7 + cols)
    rc, want = mico.wavelet_v2_compress(img, 4095, levels)
    if rc != 0:
        with pytest.raises(mic.MicError) as e:
            mic.wavelet_v2_compress(img, rows, cols, 4095, levels)
        assert e.value.code == rc
        return
    got = mic.wavelet_v2_compress(img, rows, cols, 4095, levels)
    assert got == want
    px, r, c = mic.wavelet_v2_decompress(got)
    assert (r, c) == (rows, cols) and np.array_equal(px, img)


def test_wavelet_v2_escape_path_full_16bit(mic, mico, synth, gpu_ready):
    """Coefficients beyond +-32767 take the 3-word escape (waveletfsecompressu16.go:33-37)."""
    img = synth.xr_like(cols=300, rows=200, depth=16, seed=4)
    img[50:60, 100:140] = 65535; img[60:70, 100:140] = 0
    rc, want = mico.wavelet_v2_compress(img, 65535, 5)
    assert rc == 0
    got = mic.wavelet_v2_compress(img, 200, 300, 65535, 5)
    assert got == want
    px, _, _ = mic.wavelet_v2_decompress(got)
    assert np.array_equal(px, img)


def test_frame_streams_with_damaged_tokens_agree_with_the_oracle(mic, mico, synth, gpu_ready):
    """Token streams no encoder writes (zero counts -- the reference reads one as a literal chunk of 65536 - midCount --, headers that
    point past the end, streams that stop early), put through the entropy coder again so that the FSE stage accepts them: the GPU's
    header walkers and the oracle's pull decoder (rledecompressu16.go:59-85) must make the same frame, or both an error, of each."""
    rng = np.random.default_rng(9)
    total = decodable = 0
    for seed, noise, (h, w) in ((3, 2.0, (200, 320)), (4, 60.0, (130, 257)), (5, 10.0, (64, 96))):
        img = synth.xr_like(cols=w, rows=h, depth=12, seed=seed, noise=noise)
        tok = mico.delta_rle_compress(img, 4095)
        for k in range(60):
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
                assert np.array_equal(got, want), (seed, k)
                decodable += 1
            total += 1
    assert total > 100 and decodable > 40 and total - decodable > 20


def test_wavelet_v2_mutated_streams_agree_with_the_oracle(mic, mico, synth, gpu_ready):
    """The decode tail walks the RLE headers in parts that start at arbitrary tokens (mic_wavelet.hip): whatever a damaged stream
    makes of that -- an error, or pixels -- must be what the oracle's serial decoder makes of it.  Frames big enough for several
    parts, smooth (run-dense) and noisy (literal-dense), byte flips all over the stream, truncations, a wrong symbol count."""
    rng = np.random.default_rng(5)
    checked = errors = 0
    for seed, noise, shape in ((3, 2.0, (300, 420)), (4, 40.0, (257, 389)), (5, 5.0, (96, 64))):
        img = synth.xr_like(cols=shape[1], rows=shape[0], depth=12, seed=seed, noise=noise)
        rc, good = mico.wavelet_v2_compress(img, 4095, 4)
        assert rc == 0
        px, _, _ = mic.wavelet_v2_decompress(good)
        assert np.array_equal(px, img)
        cases = [good[:n] for n in (11, 13, 20, len(good) // 2, len(good) - 1)]
        for _ in range(40):
            b = bytearray(good)
            for _ in range(int(rng.integers(1, 4))):
                i = int(rng.integers(11, len(b)))
                b[i] ^= 1 << int(rng.integers(0, 8))
            cases.append(bytes(b))
        # ... and damage to the TOKENS themselves, coded again: streams the entropy stage accepts, with headers that point anywhere
        # (zero counts, run headers at the end, literal chunks past it, a symbol count that is too large or too small)
        rc, tok = mico.fse_decompress_auto(good[11:], img.size * 4)
        assert rc == 0
        for k in range(40):
            t = tok.copy()
            for _ in range(int(rng.integers(1, 5))):
                i = int(rng.integers(0, t.size))
                t[i] = (0, 1, 2, int(t[0]), int(t[0]) // 2 + 1, int(rng.integers(0, int(t[0]) + 1)))[int(rng.integers(0, 6))] if i else t[i]
            if k % 8 == 0: t[2] = (int(t[2]) + int(rng.integers(1, 50))) & 0xFFFF          # announces more symbols than there are
            if k % 8 == 1: t[2] = max(int(t[2]) - int(rng.integers(1, 50)), 0)              # ... fewer
            if k % 8 == 2: t = t[: int(rng.integers(3, t.size))]                             # tokens end early
            rc, stream = mico.fse_compress(t, 4)
            if rc == 0:
                cases.append(good[:11] + stream)
        for c in cases:
            rc_o, want = mico.wavelet_v2_decompress(c)
            try:
                got, _, _ = mic.wavelet_v2_decompress(c)
                rc_g = 0
            except mic.MicError as e:
                rc_g, got = e.code, None
            assert (rc_g == 0) == (rc_o == 0), (seed, rc_g, rc_o, len(c))
            if rc_o == 0:
                assert np.array_equal(got, want)
            else:
                errors += 1
            checked += 1
    assert checked > 200 and errors > 20 and checked - errors > 20


# ---- gradient-adaptive predictor and PICA (deltagradrlecompressu16.go, parallelstripsadaptive.go) -------------------------
def _pica_images(synth):
    mr = np.fromfile(os.path.join(GOLDEN, "MR_256_256_image.bin"), dtype="<u2").reshape(256, 256)
    ct = np.fromfile(os.path.join(GOLDEN, "CT_512_512_image.bin"), dtype="<u2").reshape(512, 512)
    xr = synth.xr_like(cols=601, rows=403, depth=12, seed=4)
    return [("MR", mr, int(mr.max())), ("CT", ct, int(ct.max())), ("XR", xr, 4095)]


def test_single_frame_grad_matches_oracle(mic, mico, synth, gpu_ready):
    """CompressSingleFrameGrad / DecompressSingleFrameGrad: bit-exact streams, both directions, odd shapes and edge columns."""
    cases = _pica_images(synth)
    rng = np.random.default_rng(11)
    base = cases[2][1]
    for w, h in ((1, 64), (2, 40), (3, 33), (5, 70), (63, 65), (64, 64), (129, 130), (601, 1)):
        cases.append((f"{w}x{h}", np.ascontiguousarray(base[:h, :w]) if w > 8 else rng.integers(1000, 1100, size=(h, w), dtype=np.uint16), 4095))
    spikes = base.copy(); spikes[::7, ::5] = 4095; spikes[3::11, 1::9] = 0                    # escapes inside gradient neighbourhoods
    cases.append(("spikes", spikes, 4095))
    for name, img, mx in cases:
        h, w = img.shape
        rc, want = mico.compress_single_frame_grad(img, mx)
        if rc != 0:
            with pytest.raises(mic.MicError) as e:
                mic.compress_single_frame_grad(img, w, h, mx)
            assert e.value.code == rc, name
            continue
        got = mic.compress_single_frame_grad(img, w, h, mx)
        assert got == want, name
        assert np.array_equal(mic.decompress_single_frame_grad(want, w, h), img), name
    wide = synth.xr_like(cols=9000, rows=70, depth=12, seed=6)                                  # the wide row-buffer class of the predictor kernel
    rc, want = mico.compress_single_frame_grad(wide, 4095)
    assert rc == 0 and mic.compress_single_frame_grad(wide, 9000, 70, 4095) == want
    assert np.array_equal(mic.decompress_single_frame_grad(want, 9000, 70), wide)


@pytest.mark.parametrize("strips", [1, 4, 8, 16])
def test_pica_matches_oracle(mic, mico, synth, gpu_ready, strips):
    for name, img, mx in _pica_images(synth):
        h, w = img.shape
        rc, want = mico.pica_compress(img, mx, strips)
        if rc != 0:                                                                             # a strip neither predictor can code: same error
            with pytest.raises(mic.MicError) as e:
                mic.compress_parallel_strips_adaptive(img, w, h, mx, strips)
            assert e.value.code == rc, name
            continue
        got = mic.compress_parallel_strips_adaptive(img, w, h, mx, strips)
        assert got == want, name
        assert np.array_equal(mic.decompress_parallel_strips_adaptive(want), img), name


def test_pica_full_size_and_corrupt_headers(mic, mico, synth, gpu_ready):
    xr = synth.xr_like(cols=2577, rows=2048, depth=12, seed=2)
    blob = mic.compress_parallel_strips_adaptive(xr, 2577, 2048, 4095, 8)
    assert blob[:4] == b"PICA" and np.array_equal(mic.decompress_parallel_strips_adaptive(blob), xr)
    rc, back = mico.pica_decompress(blob)                                                      # the CPU restatement reads what the GPU wrote
    assert rc == 0 and np.array_equal(back, xr)
    flat = np.full((100, 37), 9, np.uint16); flat[50, 3] = 10
    rc, want = mico.pica_compress(flat, 255, 4)
    if rc == 0:
        assert mic.compress_parallel_strips_adaptive(flat, 37, 100, 255, 4) == want
    for bad in (b"PICS" + bytes(40), blob[:30], blob[:16] + bytes(16 * 8), b"PICA" + (5).to_bytes(4, "little") + (5).to_bytes(4, "little") + (0).to_bytes(4, "little")):
        with pytest.raises(mic.MicError):
            mic.decompress_parallel_strips_adaptive(bad)


# ---- device-resident sessions (what bench.py times): pixels in HBM, blobs in HBM ---------------------------------------------
def test_session_device_resident_units(mic, mico, synth, gpu_ready):
    """mic_hip_session_*: ragged units of one HBM buffer, both predictors and all state counts in ONE batch; every blob equals the
    oracle's for that unit, and the decode of the packed device blobs restores the buffer."""
    torch = pytest.importorskip("torch")
    W = 333
    img = synth.xr_like(cols=W, rows=400, depth=12, seed=21)
    specs = [(0, 64, 2), (64, 37, 4), (101, 1, 2), (102, 130, 8), (232, 100, 2 | mic.MIC_HIP_PRED_GRAD), (332, 68, 2 | mic.MIC_HIP_PRED_GRAD)]
    units = mic.Session.make_units([(y0 * W, W, hh, 4095, ns) for (y0, hh, ns) in specs])
    d_px = torch.from_numpy(img.view(np.int16).copy()).cuda()
    d_out = torch.zeros_like(d_px)
    sess = mic.Session(len(specs), W * 130)
    try:
        sess.encode_enqueue(d_px.data_ptr(), units)
        d_blobs, offs, st, used = sess.encode_finish()
        total = int(offs[-1])
        import ctypes as C
        host = np.empty(total, np.uint8)
        assert C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(d_blobs), C.c_size_t(total), 2) == 0   # D2H
        for k, (y0, hh, ns) in enumerate(specs):
            sub = img[y0: y0 + hh]
            rc, want = (mico.compress_single_frame_grad(sub, 4095) if ns & mic.MIC_HIP_PRED_GRAD else mico.compress_single_frame(sub, 4095, ns & 0xFF))
            assert st[k] == rc, (k, st[k], rc)
            if rc == 0:
                assert host[int(offs[k]): int(offs[k + 1])].tobytes() == want, k
        ok = [k for k in range(len(specs)) if st[k] == 0]
        assert len(ok) >= 4
        sess.decode_enqueue(d_blobs, offs, units, d_out.data_ptr())
        dst = sess.decode_finish()
        back = d_out.cpu().numpy().view(np.uint16)
        for k, (y0, hh, ns) in enumerate(specs):
            if st[k] == 0:
                assert dst[k] == 0 and np.array_equal(back[y0: y0 + hh], img[y0: y0 + hh]), k
        with pytest.raises(mic.MicError):                                        # an unknown flag in nstates
            sess.encode_enqueue(d_px.data_ptr(), mic.Session.make_units([(0, W, 8, 4095, 2 | 0x400)]))
    finally:
        sess.close()


def test_wavelet_batch_matches_single_calls(mic, mico, synth, gpu_ready):
    """mic_hip_wavelet_v2_{compress,decompress}_batch: every frame's file equals the oracle's single-image file; a frame that
    fails or a file of another shape is reported per frame."""
    base = synth.xr_like(cols=301, rows=203, depth=12, seed=3)
    frames = np.stack([np.roll(base, 5 * k, axis=1) + np.uint16(k) for k in range(7)])
    res = mic.wavelet_v2_compress_batch(frames, 4095 + 7, levels=4)
    files = []
    for k, (st, blob) in enumerate(res):
        rc, want = mico.wavelet_v2_compress(frames[k], 4095 + 7, 4)
        assert st == rc and (rc != 0 or blob == want), k
        files.append(want)
    sts, back = mic.wavelet_v2_decompress_batch(files)
    assert sts == [0] * 7 and np.array_equal(back, frames)
    other = mic.wavelet_v2_compress(base[:100, :150], 100, 150, 4095, 3)
    broken = bytearray(files[2]); broken[12] = 0x02                           # not a four-state stream (:503)
    sts, back = mic.wavelet_v2_decompress_batch([files[0], other, bytes(broken), files[3]])
    assert sts[0] == 0 and sts[3] == 0 and sts[1] == mic.MIC_ERR_ARGS and sts[2] != 0
    assert np.array_equal(back[0], frames[0]) and np.array_equal(back[3], frames[3])


# ---- the reference's own C codec, where its in-place build travelled with the repo (oracle/_ref, test infrastructure) -----------
def _ref_codec():
    import ctypes as C
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libmic_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libmic_ref.so not built (needs /root/reference at build time)")
    L = C.CDLL(path)
    for name in ("two", "four", "eight"):
        getattr(L, f"mic_compress_{name}_state").argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        getattr(L, f"mic_decompress_{name}_state").argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
    return L


def test_bitstreams_equal_the_reference_c_codec(mic, synth, gpu_ready):
    """Same inputs, same bytes: mic_compress_{two,four,eight}_state (ojph/mic_compress_c.c, maxValue = the frame's own maximum) against
    the library, full-size XR / CT / MR shaped frames and odd ones; each side decodes the other's stream."""
    import ctypes as C
    L = _ref_codec()
    frames = [synth.xr_like(cols=2577, rows=2048, depth=12, seed=31), synth.xr_like(cols=2577, rows=256, depth=12, seed=32),
              synth.ct_stack(frames=1, size=512, depth=12, seed=5)[0], synth.xr_like(cols=333, rows=77, depth=10, seed=33),
              np.fromfile(os.path.join(GOLDEN, "CT_512_512_image.bin"), dtype="<u2").reshape(512, 512)]
    for k, img in enumerate(frames):
        img = np.ascontiguousarray(img)
        h, w = img.shape
        mx = int(img.max())
        for name, ns in (("two", 2), ("four", 4), ("eight", 8)):
            out = np.empty(img.size * 4 + 135168, dtype=np.uint8)
            n = C.c_size_t(0)
            rc = getattr(L, f"mic_compress_{name}_state")(img.ctypes.data, w, h, out.ctypes.data, out.size, C.byref(n))
            assert rc == 0, (k, name)
            want = out[: n.value].tobytes()
            got = mic.compress_single_frame(img, w, h, mx, ns)
            assert got == want, (k, name, len(got), len(want))
            assert np.array_equal(mic.decompress_single_frame(want, w, h), img), (k, name)
            back = np.empty_like(img)
            g = np.frombuffer(got, dtype=np.uint8)
            assert getattr(L, f"mic_decompress_{name}_state")(g.ctypes.data, g.size, back.ctypes.data, w, h) == 0 and np.array_equal(back, img), (k, name)


def test_pics_files_decode_with_the_reference_c_decoder(mic, mico, synth, gpu_ready):
    """mic_decompress_parallel (ojph/mic_parallel.c:49, the reference's own C reader of PICS files, two- and four-state strips) reads
    the PICS files the library writes, scalar and SIMD inner decoders, any thread count."""
    import ctypes as C
    L = _ref_codec()
    for fn in ("mic_decompress_parallel", "mic_decompress_parallel_scalar"):
        getattr(L, fn).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int]
    for img, mx in ((synth.xr_like(cols=2577, rows=2048, depth=12, seed=41), 4095), (synth.xr_like(cols=301, rows=203, depth=12, seed=42), 4095)):
        h, w = img.shape
        for strips, ns in ((8, 2), (8, 4), (5, 2), (1, 4), (16, 2)):
            rc, want = mico.pics_compress(img, mx, strips, ns)
            if rc != 0:                                                  # strips too short for their alphabet: the reference fails too
                with pytest.raises(mic.MicError) as e:
                    mic.compress_parallel_strips(img, w, h, mx, strips, ns)
                assert e.value.code == rc
                continue
            blob = np.frombuffer(mic.compress_parallel_strips(img, w, h, mx, strips, ns), dtype=np.uint8)
            assert blob.tobytes() == want
            for fn, threads in (("mic_decompress_parallel", 0), ("mic_decompress_parallel", 3), ("mic_decompress_parallel_scalar", 8)):
                back = np.zeros_like(img)
                assert getattr(L, fn)(blob.ctypes.data, blob.size, back.ctypes.data, w, h, threads) == 0, (strips, ns, fn)
                assert np.array_equal(back, img), (strips, ns, fn)


def test_pics_and_mic2_payloads_are_reference_written_bytes_when_every_unit_holds_max_value(mic, synth, gpu_ready):
    """PICS strips and MIC2 frames are CompressSingleFrame(unit, GLOBAL maxValue) (parallelstrips.go:88, multiframecompress.go:201);
    the reference's C encoder derives maxValue from the unit it is given (ojph/mic_compress_c.c:774-775).  Where every unit holds a
    pixel equal to the global maximum the two agree, and the container payloads -- not only their readers -- are pinned to bytes the
    reference wrote: strip s of the PICS file == mic_compress_two_state(strip s), frame i of the MIC2 file likewise."""
    import ctypes as C
    L = _ref_codec()

    def ref_two_state(px):
        px = np.ascontiguousarray(px)
        out = np.empty(px.size * 4 + 135168, dtype=np.uint8)
        n = C.c_size_t(0)
        assert L.mic_compress_two_state(px.ctypes.data, px.shape[1], px.shape[0], out.ctypes.data, out.size, C.byref(n)) == 0
        return out[: n.value].tobytes()

    img = synth.xr_like(cols=2577, rows=2048, depth=12, seed=51).copy()
    img[np.arange(0, 2048, 256) + 100, 1000] = 4095                              # one pixel at the global maximum in each of the 8 strips
    assert int(img.max()) == 4095
    pics = mic.compress_parallel_strips(img, 2577, 2048, 4095, 8)
    tab = np.frombuffer(pics, dtype="<u4", count=16, offset=20).reshape(8, 2)
    for s_ in range(8):
        a = 20 + 64 + int(tab[s_, 0])
        assert pics[a: a + int(tab[s_, 1])] == ref_two_state(img[s_ * 256:(s_ + 1) * 256]), s_
    stack = synth.ct_stack(frames=12, size=512, depth=12, seed=9).copy()
    mx = int(stack.max())
    stack[:, 7, 7] = mx                                                           # every frame holds the stack's maximum
    mic2 = mic.compress_multi_frame(stack, 512, 512, mx)
    ft = np.frombuffer(mic2, dtype="<u4", count=24, offset=20).reshape(12, 2)
    for i in range(12):
        a = 20 + 96 + int(ft[i, 0])
        assert mic2[a: a + int(ft[i, 1])] == ref_two_state(stack[i]), i


def test_sub_batch_loops(gpu_ready):
    """The container calls cut long unit lists into sub-batches under a workspace ceiling (24 GiB); with MIC_HIP_WS_BUDGET_MB=8 the same
    loops run on small inputs (tests/chunking_check.py, a child process: the ceiling is read once per process)."""
    import subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MIC_HIP_WS_BUDGET_MB="8")
    r = subprocess.run([sys.executable, os.path.join(here, "chunking_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sub-batch loops ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])

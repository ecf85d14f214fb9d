"""CPU tests: pin the oracle (oracle/libmic_oracle.so) against the golden vectors produced by
the reference's own C codec, and exercise the reference tests' edge cases on it.
No GPU, no product code."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

GOLD = json.load(open(os.path.join(GOLDEN, "golden.json")))
REF_DIR = "/root/reference/testdata/compsamples_refanddir/IMAGES/REF"


def _img(name):
    if name == "CT":
        return np.fromfile(os.path.join(GOLDEN, "CT_512_512_image.bin"), dtype="<u2").reshape(512, 512)
    if name == "MR":
        return np.fromfile(os.path.join(GOLDEN, "MR_256_256_image.bin"), dtype="<u2").reshape(256, 256)
    path = f"{REF_DIR}/{name}_UNC"
    if not os.path.exists(path):
        pytest.skip("NEMA inputs only exist in the build container")
    b = open(path, "rb").read()
    i = b.rfind(bytes([0xE0, 0x7F, 0x10, 0x00]))
    ln = int.from_bytes(b[i + 8:i + 12], "little")
    rec = GOLD["streams"][f"{name}/2"]
    w, h = rec["width"], rec["height"]
    return np.frombuffer(b[i + 12:i + 12 + ln], dtype="<u2")[: w * h].reshape(h, w).copy()


@pytest.mark.parametrize("key", sorted(GOLD["streams"].keys()))
def test_oracle_matches_reference_c_stream(mico, key):
    """Byte parity of CompressSingleFrame{,4State,8State} with the reference C encoder
    (which equals Go when maxValue == image max)."""
    rec = GOLD["streams"][key]
    name = key.split("/")[0]
    img = _img(name)
    assert f"{mico.fnv1a64(img.tobytes()):016x}" == rec["pixels_fnv1a64"]
    rc, blob = mico.compress_single_frame(img, rec["max_value"], rec["nstates"])
    assert rc == 0
    assert len(blob) == rec["size"]
    assert blob[:32].hex() == rec["head"] and blob[-32:].hex() == rec["tail"]
    assert f"{mico.fnv1a64(blob):016x}" == rec["fnv1a64"]
    rc, px = mico.decompress_single_frame(blob, rec["width"], rec["height"])
    assert rc == 0 and np.array_equal(px, img)


@pytest.mark.parametrize("ns", [2, 4, 8])
def test_oracle_decodes_reference_c_mr_stream(mico, ns):
    blob = open(os.path.join(GOLDEN, f"MR_256_256_{ns}state.mic"), "rb").read()
    rc, px = mico.decompress_single_frame(blob, 256, 256)
    assert rc == 0 and np.array_equal(px, _img("MR"))
    assert blob[:2] == bytes([0xFF, {2: 0x02, 4: 0x04, 8: 0x84}[ns]])   # fse2state_test.go:117-142


# ---- edge cases of the reference's FSE tests (fse2state_test.go:178-257, fse4state_test.go:105-,
# ---- fse8state_test.go:106-, rans8state_test.go:105-147)
@pytest.mark.parametrize("ns", [1, 2, 4, 8, 108])
def test_fse_all_same_is_use_rle(mico, ns):
    rc, _ = mico.fse_compress(np.full(1000, 42, dtype=np.uint16), ns)
    assert rc == mico.ERR_USE_RLE


@pytest.mark.parametrize("ns", [1, 2, 4, 8, 108])
def test_fse_two_elements_error(mico, ns):
    rc, _ = mico.fse_compress(np.array([1, 2], dtype=np.uint16), ns)
    assert rc != 0


@pytest.mark.parametrize("ns", [1, 2, 4, 8, 108])
@pytest.mark.parametrize("n", list(range(9, 16)) + list(range(101, 108)) + list(range(1001, 1008)))
def test_fse_ragged_lengths_round_trip(mico, ns, n):
    for mod in (8, 17):
        data = (np.arange(n) % mod).astype(np.uint16)
        rc, blob = mico.fse_compress(data, ns)
        if rc != 0:
            assert rc == mico.ERR_INCOMPRESSIBLE
            continue
        rc, out = mico.fse_decompress_auto(blob, n + 16)
        assert rc == 0 and np.array_equal(out, data)


def test_incompressible_noise_falls_through(mico, synth):
    # 2^21 uniform 16-bit symbols: maxCount (~60) < n>>15 (64) -> ErrIncompressible (fse2state.go:40)
    noise = (synth.hash_u64(1 << 21, 9) & np.uint64(0xFFFF)).astype(np.uint16)
    for ns in (1, 2, 4, 8, 108):
        rc, _ = mico.fse_compress(noise, ns)
        assert rc == mico.ERR_INCOMPRESSIBLE


def test_tiny_noisy_input_reports_internal_error(mico, synth):
    """More distinct symbols than table slots: the Go normaliser spins forever
    (fsecompressu16.go:627-635); the restatement reports an error instead of hanging."""
    noise = (synth.hash_u64(4096, 9) & np.uint64(0xFFFF)).astype(np.uint16)
    rc, _ = mico.fse_compress(noise, 2)
    assert rc not in (0, mico.ERR_USE_RLE)


# ---- PICS (parallelstrips_test.go:17-50, :82-145)
@pytest.mark.parametrize("strips", [1, 2, 4, 8])
@pytest.mark.parametrize("ns", [2, 4, 8])
def test_pics_round_trip(mico, strips, ns):
    img = _img("MR")
    rc, blob = mico.pics_compress(img, int(img.max()), strips, ns)
    assert rc == 0 and blob[:4] == b"PICS"
    rc, px = mico.pics_decompress(blob)
    assert rc == 0 and np.array_equal(px, img)


def test_pics_ct_ratio_matches_published(mico):
    """Published PICS-8 ratio for CT is 1.962 (results/20260518-054951/02-all-codecs-encode.txt)."""
    img = _img("CT")
    rc, blob = mico.pics_compress(img, int(img.max()), 8, 2)
    assert rc == 0
    assert abs(img.size * 2 / len(blob) - 1.962) < 0.002


def test_pics_strips_clamped_to_height(mico):
    full = _img("MR")
    img = full[:2].copy()                                      # parallelstrips_test.go:119-145
    rc, blob = mico.pics_compress(img, int(full.max()), 256, 2)
    assert rc == 0
    assert int.from_bytes(blob[12:16], "little") == 2
    rc, px = mico.pics_decompress(blob)
    assert rc == 0 and np.array_equal(px, img)


def test_pics_bad_magic_and_truncation(mico):
    img = _img("MR")
    rc, blob = mico.pics_compress(img, int(img.max()), 4, 2)
    rc, _ = mico.pics_decompress(b"XXXX" + blob[4:])
    assert rc == mico.ERR_CORRUPT
    rc, _ = mico.pics_decompress(blob[:10])
    assert rc == mico.ERR_CORRUPT
    rc, _ = mico.pics_decompress(blob[: len(blob) // 2])
    assert rc != 0


# ---- MIC2 (multiframe_test.go:149-243)
@pytest.mark.parametrize("temporal", [False, True])
def test_mic2_round_trip(mico, synth, temporal):
    stack = synth.ct_stack(frames=5, size=128, depth=12, seed=11)
    rc, blob = mico.mic2_compress(stack, 4095, temporal)
    assert rc == 0 and blob[:4] == b"MIC2"
    assert blob[16] == (0x03 if temporal else 0x01)
    rc, out = mico.mic2_decompress(blob)
    assert rc == 0 and np.array_equal(out, stack)


def test_delta_symbol_stream_shape(mico):
    """First stream word is the delimiter, first symbol is maxValue (deltarlecompressu16.go:25-29)."""
    img = _img("MR")
    mx = int(img.max())
    tok = mico.delta_rle_compress(img, mx)
    depth = mx.bit_length()
    assert tok[0] == (1 << depth) - 1
    sym = mico.delta_symbols(img, mx)
    assert sym[0] == mx and sym.size >= img.size + 1

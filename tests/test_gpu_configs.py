"""GPU tests at the sizes BASELINE.json names (configs 3, 4 and 5; config 2, PICS-8 on the 2577 x 2048 XR shape, is in
test_gpu_parity.py).  Everything goes through the C ABI and is compared with the CPU oracle on the same seeded inputs;
the 32768 x 32768 slide is checked against the oracle on the parts the oracle does in seconds (an 8192 x 8192 slide in full,
sampled tiles of the big one byte for byte) and through the round trip for the rest."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


# ---- config 3: WaveletV2SIMDRLEFSECompressU16, 5 levels, CR shape rows 2140 x cols 1760 (waveletfsecompressu16.go:427) ----
def test_config3_wavelet_v2_on_cr_shape(mic, mico, synth, gpu_ready):
    img = synth.cr_like()                                            # cols 1760, rows 2140 (fseu16_test.go:31)
    rows, cols = img.shape
    assert (rows, cols) == (2140, 1760)
    rc, want = mico.wavelet_v2_compress(img, 4095, 5)
    assert rc == 0
    got = mic.wavelet_v2_compress(img, rows, cols, 4095, 5)
    assert got == want
    assert got[10] == 5 and got[11:13] == bytes([0xFF, 0x04])        # 5 levels applied, 4-state FSE, no fallback
    px, r, c = mic.wavelet_v2_decompress(got)
    assert (r, c) == (rows, cols) and np.array_equal(px, img)
    rc, opx = mico.wavelet_v2_decompress(got)                        # and the oracle reads the device's file
    assert rc == 0 and np.array_equal(opx, img)


def test_config3_wavelet_v2_batch_of_cr_frames(mic, mico, synth, gpu_ready):
    """Several CR-shaped frames side by side (the batch entry points): every file equals the oracle's for that frame."""
    frames = np.stack([synth.xr_like(cols=1760, rows=2140, depth=12, seed=20 + i, noise=5.0 + 3.0 * i) for i in range(3)])
    res = mic.wavelet_v2_compress_batch(frames, 4095, 5)
    files = []
    for i, (st, blob) in enumerate(res):
        rc, want = mico.wavelet_v2_compress(frames[i], 4095, 5)
        assert st == 0 and rc == 0 and blob == want
        files.append(blob)
    st, px = mic.wavelet_v2_decompress_batch(files)
    assert st == [0, 0, 0] and np.array_equal(px.reshape(frames.shape), frames)


# ---- config 4: MIC2 independent mode, 512 frames of 512 x 512 (multiframecompress.go:179) ---------------------------------
def test_config4_mic2_independent_512_cubed(mic, mico, synth, gpu_ready):
    stack = synth.ct_stack(512, 512, 12, seed=3)
    rc, want = mico.mic2_compress(stack, 4095, False)
    assert rc == 0
    got = mic.compress_multi_frame(stack, 512, 512, 4095, temporal=False)
    assert got == want
    assert got[:4] == b"MIC2" and int.from_bytes(got[12:16], "little") == 512 and got[16] == 0x01
    back = mic.decompress_multi_frame(got)
    assert back.shape == stack.shape and np.array_equal(back, stack)
    for idx in (0, 255, 511):                                        # DecompressFrame, multiframecompress.go:266
        assert np.array_equal(mic.decompress_frame(got, idx), stack[idx])


# ---- config 5: MIC3, 8-bit RGB, 256 x 256 tiles (wsicompress.go:27) ---------------------------------------------------
def test_config5_wsi_8192_matches_oracle(mic, mico, synth, gpu_ready):
    """A slide the oracle codes in a few seconds, in full: 6 pyramid levels, 1365 tiles."""
    W = H = 8192
    slide = synth.wsi_slide(W, H, seed=4)
    rc, want = mico.wsi_compress(slide)
    assert rc == 0
    got = mic.compress_wsi(slide, W, H)
    assert got == want
    hdr = mic.read_wsi_header(got)
    assert len(hdr["levels"]) == 6 and hdr["total_tiles"] == 1024 + 256 + 64 + 16 + 4 + 1
    assert np.array_equal(mic.decompress_wsi_level(got, 0), slide)


def _box2(a):
    """Downsample2xRGB (wsipyramid.go:10-32)"""
    h, w = a.shape[0] // 2 * 2, a.shape[1] // 2 * 2
    b = a[:h, :w].astype(np.uint16)
    return ((b[0::2, 0::2] + b[0::2, 1::2] + b[1::2, 0::2] + b[1::2, 1::2] + 2) // 4).astype(np.uint8)


def test_config5_wsi_32768_slide(mic, mico, synth, gpu_ready):
    """The configuration's own size: 32768 x 32768 RGB, 8 levels, 21 845 tiles, 65 535 planes.  The file's
    layout, the level-0 round trip, and -- against the oracle -- the bytes of sampled tile blobs on levels 0-4 (each a pure
    function of its source pixels: compressRGBTileBlob, wsicompress.go:319-363) and the pixels of sampled tiles on every level."""
    W = H = 32768
    slide = synth.wsi_slide(W, H, seed=4, workers=16)
    blob = mic.compress_wsi(slide, W, H)
    hdr = mic.read_wsi_header(blob)
    dims = [(l["width"], l["height"], l["tiles_x"], l["tiles_y"]) for l in hdr["levels"]]
    assert dims == [(W >> k, H >> k, max(1, (W >> k) // 256), max(1, (H >> k) // 256)) for k in range(8)]
    assert hdr["total_tiles"] == 21845 and (hdr["tile_width"], hdr["tile_height"]) == (256, 256)
    c = np.frombuffer(blob, dtype=np.uint8)
    table = 48 + 20 * 8
    data_off = table + 16 * 21845
    ent = c[table:data_off].view("<u8").reshape(21845, 2)
    assert ent[0, 0] == 0 and np.array_equal(ent[1:, 0], np.cumsum(ent[:-1, 1]))       # blobs back to back, wsiformat.go:139-160
    assert data_off + int(ent[-1, 0] + ent[-1, 1]) == len(blob)
    first = [0, 16384, 20480, 21504, 21760, 21824, 21840, 21844]
    rng = np.random.default_rng(7)
    # byte parity of sampled tiles, levels 0-4: the level-k tile (tx, ty) is the 2^k-fold box reduction of a source crop
    for k, count in ((0, 24), (1, 12), (2, 8), (3, 6), (4, 4)):
        n = dims[k][2]
        picks = {(int(rng.integers(n)), int(rng.integers(n))) for _ in range(count)} | {(n * 35 // 100, n * 40 // 100), (0, 0)}
        for tx, ty in sorted(picks):
            s = 256 << k
            crop = slide[ty * s:(ty + 1) * s, tx * s:(tx + 1) * s]
            for _ in range(k):
                crop = _box2(crop)
            rc, want = mico.wsi_compress_tile(np.ascontiguousarray(crop))
            assert rc == 0
            gi = first[k] + ty * n + tx
            o, ln = int(ent[gi, 0]), int(ent[gi, 1])
            assert bytes(c[data_off + o: data_off + o + ln]) == want, f"level {k} tile ({tx},{ty})"
    # pixels of sampled tiles on every level, oracle reader against the device reader
    for k in range(8):
        n = dims[k][2]
        for tx, ty in {(0, 0), (n - 1, n - 1), (n * 35 // 100, n * 40 // 100)}:
            rc, t = mico.wsi_decompress_tile_at(blob, k, tx, ty)
            assert rc == 0 and np.array_equal(mic.decompress_wsi_tile(blob, k, tx, ty), t)
    lv0 = mic.decompress_wsi_level(blob, 0)
    assert lv0.shape == slide.shape and np.array_equal(lv0, slide)


# ---- the device-resident forms bench.py times (inputs, streams and pixels stay in HBM) ------------------------------------
def test_session_wavelet_v2_device_resident(mic, mico, synth, gpu_ready):
    torch = pytest.importorskip("torch")
    frames = np.stack([synth.xr_like(cols=1760, rows=2140, depth=12, seed=30 + i, noise=5.0) for i in range(2)])
    d_px = torch.from_numpy(frames.view(np.int16).copy()).cuda()
    d_out = torch.zeros_like(d_px)
    sess = mic.Session(2, 2 * 2140 * 1760 + 16, device=0)               # mic_hip_session_create_on
    assert sess.device == 0
    d_streams, offs, st, applied = sess.wavelet_v2_encode(d_px.data_ptr(), 2, 2140, 1760, 5)
    assert (st == 0).all() and applied == 5
    packed = torch.empty(int(offs[-1]), dtype=torch.uint8, device="cuda")
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(ctypes.c_void_p(packed.data_ptr()), ctypes.c_void_p(d_streams), ctypes.c_size_t(int(offs[-1])), 3) == 0   # device to device
    host = packed.cpu().numpy().tobytes()
    for i in range(2):
        rc, want = mico.wavelet_v2_compress(frames[i], 4095, 5)
        hdr = (2140).to_bytes(4, "little") + (1760).to_bytes(4, "little") + (4095).to_bytes(2, "little") + bytes([applied])
        assert rc == 0 and hdr + host[int(offs[i]):int(offs[i + 1])] == want
    st = sess.wavelet_v2_decode(packed.data_ptr(), offs, 2, 2140, 1760, applied, d_out.data_ptr())
    assert (st == 0).all() and torch.equal(d_out, d_px)
    sess.close()


def test_session_wsi_device_resident(mic, mico, synth, gpu_ready):
    torch = pytest.importorskip("torch")
    W, H = 2304, 1800                                                    # 4 levels, ragged edge tiles, white + tissue + constant planes
    slide = synth.wsi_slide(W, H, seed=11)
    d_px = torch.from_numpy(slide).cuda()
    sess = mic.Session(1, 256 * 256)
    tiles, nbytes = sess.wsi_encode(d_px.data_ptr(), W, H)
    rc, want = mico.wsi_compress(slide)
    assert rc == 0 and nbytes == len(want)
    assert sess.wsi_write() == want                                      # the container around the device-resident planes
    assert mic.compress_wsi(slide, W, H) == want
    lv = sess.wsi_levels()
    assert lv[0] == (W, H) and tiles == sum(((w + 255) // 256) * ((h + 255) // 256) for w, h in lv)
    img = slide
    for k, (w, h) in enumerate(lv):
        d_out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
        sess.wsi_decode_level(k, d_out.data_ptr(), d_out.numel())
        assert np.array_equal(d_out.cpu().numpy(), img), f"level {k}"
        img = _box2(img)
    sess.close()


def test_wavelet_v2_streams_that_take_64_dwords_per_chunk(mic, mico, synth, gpu_ready):
    """tableLog-16 streams can take 16 bits a symbol: 128 symbols = 64 dwords, more than the bit-window ring holds ahead of a chunk
    unless it is refreshed mid-chunk (mic_decode_ls.hip).  Frame 2 of this batch is one (it decoded as corrupt before the refresh)."""
    torch = pytest.importorskip("torch")
    d_px = synth.xr_like_batch_torch(4, cols=1760, rows=2140, depth=12, seed0=2000, noise=5.0, device="cuda")
    d_out = torch.zeros_like(d_px)
    sess = mic.Session(4, 2 * 2140 * 1760 + 16)
    d_streams, offs, st, applied = sess.wavelet_v2_encode(d_px.data_ptr(), 4, 2140, 1760, 5)
    assert (st == 0).all()
    dst = sess.wavelet_v2_decode(d_streams, offs, 4, 2140, 1760, applied, d_out.data_ptr())
    assert (dst == 0).all() and torch.equal(d_out, d_px)
    frame2 = d_px[2].cpu().numpy().view(np.uint16)
    rc, want = mico.wavelet_v2_compress(frame2, 4095, 5)
    host = torch.empty(int(offs[3] - offs[2]), dtype=torch.uint8, device="cuda")
    mic.device_copy(host.data_ptr(), d_streams + int(offs[2]), host.numel())
    assert rc == 0 and want[11:] == host.cpu().numpy().tobytes()
    sess.close()


def test_xr_batch_generator_on_the_device_equals_numpy(synth, gpu_ready):
    """bench.py's frames are made on the device: the same hash and the same float64 operations as synth.xr_like, bit for bit."""
    pytest.importorskip("torch")
    d = synth.xr_like_batch_torch(3, cols=2577, rows=2048, depth=12, seed0=7, noise=synth.XR_NOISE_PUBLISHED_RATIO, device="cuda")
    want = synth.xr_like(cols=2577, rows=2048, depth=12, seed=8, noise=synth.XR_NOISE_PUBLISHED_RATIO)
    assert np.array_equal(d[1].cpu().numpy().view(np.uint16), want)


def test_parallel_gather_on_rccl_world_size_1(mic, mico, synth, gpu_ready):
    """parallel.py on the backend bench.py uses (`nccl` = RCCL) with the real session codec and device tensors: one rank here (the
    box has one GPU); the two-rank logic runs under gloo in tests/test_parallel_cpu.py."""
    torch = pytest.importorskip("torch")
    import importlib
    import torch.distributed as dist
    par = importlib.import_module("medical_image_codec_amd.parallel")
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(29500 + os.getpid() % 2000)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        stack = np.stack([synth.xr_like(cols=320, rows=200, depth=12, seed=60 + i) for i in range(6)])
        d_px = torch.from_numpy(stack.view(np.int16).copy()).cuda()
        sess = mic.Session(12, 320 * 200, device=0)
        units = [(i * 320 * 200, 320, 200, 4095, 2) for i in range(6)]
        enc, dec = par.session_codec(mic, sess, d_px, units)
        mic2 = par.dist_compress_multi_frame(enc, 320, 200, 6)
        rc, want = mico.mic2_compress(stack, 4095, False)
        assert rc == 0 and mic2 == want
        lo, hi, px = par.dist_decompress_multi_frame(dec, mic2, device=torch.device("cuda:0"))
        assert (lo, hi) == (0, 6) and torch.equal(px, d_px.reshape(6, 200, 320))
        sh = 100
        strips = [(f * 320 * 200 + y0 * 320, 320, sh, 4095, 2) for f in range(6) for y0 in (0, 100)]
        enc_s, _ = par.session_codec(mic, sess, d_px, strips)
        pics = par.dist_compress_pics_batch(enc_s, 320, 200, 2, 6)
        for f in range(6):
            rc, pw = mico.pics_compress(stack[f], 4095, 2, 2)
            assert rc == 0 and pics[f] == pw
        sess.close()
        # MIC3 in bands: with one rank the band is the slide; a two-rank run's bands (tests/test_parallel_cpu.py: rows 0..1024 and
        # 1024..1300 of this slide, levels 0..2) must come out of the session as they come out of the oracle
        slide = synth.wsi_slide(520, 1300, seed=11, workers=1)
        d_slide = torch.from_numpy(slide).cuda()
        ws = mic.Session(64, 256 * 256, device=0)
        enc_slide = par.session_wsi_codec(mic, ws)
        rc, want3 = mico.wsi_compress(slide)
        assert rc == 0 and par.dist_compress_wsi(enc_slide, d_slide, 520, 1300) == want3
        def as_file(img, levels):                                     # the band's container around the device payload the codec hands on
            payload, sz, lv = enc_slide(img, levels)
            assert payload.is_cuda
            return par.mic3_header(int(img.shape[1]), int(img.shape[0]), 256, 256, 3, 8, lv, sz) + payload.cpu().numpy().tobytes()
        for y0, y1 in ((0, 1024), (1024, 1300)):
            rc, wb = mico.wsi_compress(np.ascontiguousarray(slide[y0:y1]), 256, 256, 3)
            assert rc == 0 and as_file(d_slide[y0:y1], 3) == wb
        top = par.downsample2x(par.downsample2x(par.downsample2x(d_slide)))
        rc, wt = mico.wsi_compress(np.ascontiguousarray(top.cpu().numpy()), 256, 256, 1)
        assert rc == 0 and as_file(top, 1) == wt
        ws.close()
    finally:
        dist.destroy_process_group()

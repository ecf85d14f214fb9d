"""The several-devices path of the C ABI (mic_hip_set_devices): the batch entry points cut their jobs into one contiguous shard per
listed device and run the shards side by side, each on a session of its device's pool (csrc/mic_host_io.hip: run_shards).  The test box
has one GPU, so the lists here are {0}, {0, 0} and {0, 0, 0}: the same code path with two and three pools' worth of sessions on one
device.  Every result is compared with the oracle; reference fan-outs: parallelstrips.go:77-93, multiframecompress.go:186-209."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mixed_images(synth):
    shapes = [(322, 256), (257, 200), (640, 130), (129, 77), (322, 256), (500, 64), (1100, 40), (322, 256), (96, 300), (2577, 24), (322, 256)]
    return [synth.xr_like(cols=w, rows=h, depth=12, seed=300 + i) for i, (w, h) in enumerate(shapes)]


@pytest.fixture
def device_lists(mic, gpu_ready):
    yield ([0], [0, 0], [0, 0, 0])
    mic.set_devices([0])


def test_batches_over_device_lists_equal_the_oracle(mic, mico, synth, device_lists):
    imgs = _mixed_images(synth)
    want_pics = [mico.pics_compress(im, 4095, 8, 2) for im in imgs]
    want_frames = [mico.compress_single_frame(im, 4095, 4) for im in imgs]
    for devs in device_lists:
        mic.set_devices(devs)
        assert mic.get_devices() == devs
        res = mic.compress_parallel_strips_batch(imgs, 4095, 8, 2)
        files, ok = [], []
        for im, (st, blob), (rc, want) in zip(imgs, res, want_pics):
            assert st == rc, devs                     # (thin strips of noisy frames are streams the reference cannot code either)
            if rc == 0:
                assert blob.tobytes() == want, devs
                files.append(want); ok.append(im)
        assert len(ok) >= 8
        back = mic.decompress_parallel_strips_batch(files, [(im.shape[1], im.shape[0]) for im in ok])
        for im, (st, px) in zip(ok, back):
            assert st == 0 and np.array_equal(px, im), devs
        jobs = mic.compress_batch(imgs, [4095] * len(imgs), 4)
        blobs, ok = [], []
        for im, (st, blob, used), (rc, want) in zip(imgs, jobs, want_frames):
            assert st == rc, devs
            if rc == 0:
                assert blob == want, devs
                blobs.append(want); ok.append(im)
        assert len(ok) >= 8
        back = mic.decompress_batch(blobs, [(im.shape[1], im.shape[0]) for im in ok])
        for im, (st, px) in zip(ok, back):
            assert st == 0 and np.array_equal(px, im), devs


def test_a_failing_job_stays_alone_on_its_shard(mic, mico, synth, device_lists):
    """The job the reference cannot code (10-bit noise in 1290-pixel strips: more distinct symbols than table slots) fails, whichever
    shard it lands on; its neighbours -- on the same shard and on the others -- do not."""
    good = [synth.xr_like(cols=322, rows=256, depth=12, seed=400 + i) for i in range(6)]
    bad = synth.xr_like(cols=129, rows=77, depth=10, seed=7)
    for devs in device_lists:
        mic.set_devices(devs)
        for pos in (0, 3, 6):
            imgs = good[:pos] + [bad] + good[pos:]
            res = mic.compress_parallel_strips_batch(imgs, 4095, 8, 2)
            for i, (im, (st, blob)) in enumerate(zip(imgs, res)):
                rc, want = mico.pics_compress(im, 4095, 8, 2)
                assert st == rc, (devs, pos, i)
                assert (rc != 0) == (i == pos)
                if rc == 0:
                    assert blob.tobytes() == want


def test_mic2_frames_are_sharded_and_moved_into_place(mic, mico, synth, device_lists):
    """One group, many frames: every shard codes its frames into a provisional place of the caller's buffer and the host moves them
    down (mic_hip_mic2_compress); decode shards the frames freely."""
    frames = np.stack([synth.xr_like(cols=160, rows=120, depth=12, seed=500 + i, noise=3.0 + i) for i in range(26)])
    rc, want = mico.mic2_compress(frames, 4095)
    assert rc == 0
    for devs in device_lists:
        mic.set_devices(devs)
        got = mic.compress_multi_frame(frames, 160, 120, 4095)
        assert got == want, devs
        back = mic.decompress_multi_frame(want)
        assert np.array_equal(back.reshape(frames.shape), frames), devs


def test_the_failing_strip_is_named(mic, mico, synth, gpu_ready):
    """parallelstrips.go:97 wraps a strip's error with its index ("parallelstrips: strip %d: %w"): strip 3 of eight is noise the
    entropy stage cannot code, the call fails with that strip's error and says which one it was."""
    w, sh = 322, 32
    img = synth.xr_like(cols=w, rows=8 * sh, depth=12, seed=11)
    rng = np.random.default_rng(5)
    img[3 * sh: 4 * sh, :] = rng.integers(0, 4096, (sh, w)).astype(np.uint16)
    assert [mico.compress_single_frame(img[k * sh: (k + 1) * sh], 4095, 2)[0] == 0 for k in range(8)] == [True, True, True, False] + [True] * 4
    rc, _ = mico.pics_compress(img, 4095, 8, 2)
    assert rc != 0
    with pytest.raises(mic.MicError) as e:
        mic.compress_parallel_strips(img, w, 8 * sh, 4095, 8)
    assert e.value.code == rc and e.value.strip == 3 and "strip 3" in str(e.value)
    res = mic.compress_parallel_strips_batch([img, synth.xr_like(cols=w, rows=8 * sh, depth=12, seed=12)], 4095, 8, 2)
    assert res[0][0] == rc and res[1][0] == 0
    assert mic.compress_parallel_strips_batch.failed_strips == [3, -1]
    # decode: a file whose fifth strip is cut short inside its entropy stream
    good = synth.xr_like(cols=322, rows=256, depth=12, seed=13)
    rc, f = mico.pics_compress(good, 4095, 8, 2)
    b = bytearray(f)
    off = 20 + 8 * 8 + int.from_bytes(f[20 + 8 * 4: 24 + 8 * 4], "little")
    ln = int.from_bytes(f[24 + 8 * 4: 28 + 8 * 4], "little")
    b[off + ln - 1] = 0                                              # a zero last byte: the bit reader has no end mark (bitreader.go:36-38)
    with pytest.raises(mic.MicError) as e:
        mic.decompress_parallel_strips(bytes(b))
    assert e.value.strip == 4

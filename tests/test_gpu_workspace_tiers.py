"""GPU tests of the two workspace tiers (mic_session.h): a session starts with slabs sized for ordinary frames (tier 1) and runs a
batch again in worst-case slabs (tier 2) when a unit reports it needs them -- more tokens than pixels + an eighth (a frame where most
pixels escape, deltarlecompressu16.go:52-56), or an alphabet past 8192 symbols (depth 14 and up, fseu16.go:57-58).  The retry must
not change a byte, in either direction, on the session entry points or on the host-pointer ones."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _escape_frame(w=640, h=480, maxv=4095):
    """Every second column at the top of the range: nearly every pixel is an escape (delimiter + raw value = two tokens a pixel)"""
    img = np.zeros((h, w), dtype=np.uint16)
    img[:, ::2] = maxv
    img[::7, ::3] = maxv // 3
    return img


def test_the_tier_is_sticky_but_not_for_ever(mic, mico, synth, gpu_ready):
    """VERDICT r3 (what's weak 9): one escape-heavy batch sends a session to the worst-case slabs; eight ordinary batches in a row later
    it is back in the small ones and has returned the memory -- and the streams are the oracle's all along."""
    torch = pytest.importorskip("torch")
    w, h, maxv = 640, 480, 4095
    plain = [synth.xr_like(cols=w, rows=h, depth=12, seed=70 + i) for i in range(3)]
    want = [mico.compress_single_frame(img, maxv, 2)[1] for img in plain]
    sess = mic.Session(3, w * h)
    try:
        _session_round_trip(mic, sess, plain, maxv, torch)
        small_bytes, big = sess.workspace_bytes()
        assert not big
        _session_round_trip(mic, sess, [plain[0], _escape_frame(w, h, maxv), plain[1]], maxv, torch)
        big_bytes, big = sess.workspace_bytes()
        assert big and big_bytes > 2 * small_bytes
        for k in range(9):                                              # eight calm batches un-stick it; the ninth is laid out small again
            _, host, offs, st = _session_round_trip(mic, sess, plain, maxv, torch)
            for i in range(3):
                assert st[i] == 0 and host[int(offs[i]):int(offs[i + 1])].tobytes() == want[i], (k, i)
        now_bytes, big = sess.workspace_bytes()
        assert not big and now_bytes < big_bytes // 2
        # ... and an escape-heavy batch is still coded (it asks for the large slabs again)
        mixed = [plain[0], _escape_frame(w, h, maxv), plain[1]]
        _, host, offs, st = _session_round_trip(mic, sess, mixed, maxv, torch)
        for i, img in enumerate(mixed):
            rc, f = mico.compress_single_frame(img, maxv, 2)
            assert st[i] == rc == 0 and host[int(offs[i]):int(offs[i + 1])].tobytes() == f
        assert sess.workspace_bytes()[1]
    finally:
        sess.close()


def _session_round_trip(mic, sess, imgs, maxv, torch):
    w, h = imgs[0].shape[1], imgs[0].shape[0]
    stack = np.stack(imgs)
    units = mic.Session.make_units([(i * w * h, w, h, maxv, 2) for i in range(len(imgs))])
    d_px = torch.from_numpy(stack.view(np.int16).copy()).cuda()
    sess.encode_enqueue(d_px.data_ptr(), units)
    d_blobs, offs, st, _ = sess.encode_finish()
    host = np.empty(int(offs[-1]), np.uint8)
    assert C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(d_blobs), C.c_size_t(host.size), 2) == 0
    return units, host, offs, st


def test_escape_heavy_batch_runs_again_in_tier_two_with_the_same_bytes(mic, mico, synth, gpu_ready):
    torch = pytest.importorskip("torch")
    w, h, maxv = 640, 480, 4095
    plain = [synth.xr_like(cols=w, rows=h, depth=12, seed=40 + i) for i in range(3)]
    sess = mic.Session(4, w * h)
    try:
        nbytes, big = sess.workspace_bytes()
        assert not big and nbytes <= 8 * 4 * w * h * 2                           # tier 1: under eight times the pixels it holds
        _, host, offs, st = _session_round_trip(mic, sess, plain, maxv, torch)
        assert not sess.workspace_bytes()[1]                                     # ordinary frames stay in tier 1
        for k, img in enumerate(plain):
            rc, want = mico.compress_single_frame(img, maxv, 2)
            assert st[k] == rc == 0 and host[int(offs[k]):int(offs[k + 1])].tobytes() == want
        mixed = [plain[0], _escape_frame(w, h, maxv), plain[1]]
        units, host, offs, st = _session_round_trip(mic, sess, mixed, maxv, torch)
        assert sess.workspace_bytes()[1]                                         # the escape frame did not fit: the batch ran in tier 2
        for k, img in enumerate(mixed):
            rc, want = mico.compress_single_frame(img, maxv, 2)
            assert st[k] == rc == 0 and host[int(offs[k]):int(offs[k + 1])].tobytes() == want, k
    finally:
        sess.close()
    # decode of the same streams in a fresh session: tier 1 first, the long token stream asks for tier 2
    fresh = mic.Session(3, w * h)
    try:
        assert not fresh.workspace_bytes()[1]
        d_in = torch.from_numpy(host.copy()).cuda()
        d_out = torch.zeros(3 * w * h, dtype=torch.int16, device="cuda")
        fresh.decode_enqueue(d_in.data_ptr(), offs, units, d_out.data_ptr())
        st = fresh.decode_finish()
        assert fresh.workspace_bytes()[1]
        back = d_out.cpu().numpy().view(np.uint16).reshape(3, h, w)
        for k, img in enumerate(mixed):
            assert st[k] == 0 and np.array_equal(back[k], img), k
    finally:
        fresh.close()


def test_sixteen_bit_streams_start_in_tier_two_on_encode_and_grow_on_decode(mic, mico, synth, gpu_ready):
    torch = pytest.importorskip("torch")
    w, h, maxv = 512, 300, 65535
    imgs = [synth.xr_like(cols=w, rows=h, depth=16, seed=60 + i) for i in range(2)]
    sess = mic.Session(2, w * h)
    try:
        units, host, offs, st = _session_round_trip(mic, sess, imgs, maxv, torch)
        for k, img in enumerate(imgs):
            rc, want = mico.compress_single_frame(img, maxv, 2)
            assert st[k] == rc == 0 and host[int(offs[k]):int(offs[k + 1])].tobytes() == want
    finally:
        sess.close()
    fresh = mic.Session(2, w * h)
    try:
        d_in = torch.from_numpy(host.copy()).cuda()
        d_out = torch.zeros(2 * w * h, dtype=torch.int16, device="cuda")
        fresh.decode_enqueue(d_in.data_ptr(), offs, units, d_out.data_ptr())
        st = fresh.decode_finish()
        assert fresh.workspace_bytes()[1]                                        # (a 65536-symbol table does not fit tier 1's 8192 slots)
        back = d_out.cpu().numpy().view(np.uint16).reshape(2, h, w)
        for k, img in enumerate(imgs):
            assert st[k] == 0 and np.array_equal(back[k], img)
    finally:
        fresh.close()


def test_host_entry_points_grow_without_the_caller_seeing_it(mic, mico, synth, gpu_ready):
    """PICS batch over host pointers: an escape-heavy image among ordinary ones, then a 16-bit one; bytes as the oracle writes them"""
    w, h = 640, 480
    imgs = [synth.xr_like(cols=w, rows=h, depth=12, seed=70), _escape_frame(w, h, 4095), synth.xr_like(cols=w, rows=h, depth=12, seed=71)]
    res = mic.compress_parallel_strips_batch(imgs, 4095, 4, 2)
    files = []
    for img, (st, blob) in zip(imgs, res):
        rc, want = mico.pics_compress(img, 4095, 4, 2)
        assert st == rc == 0 and blob.tobytes() == want
        files.append(want)
    out = mic.decompress_parallel_strips_batch(files, [(w, h)] * 3)
    for img, (st, px) in zip(imgs, out):
        assert st == 0 and np.array_equal(px, img)
    ct = synth.xr_like(cols=w, rows=h, depth=16, seed=72)
    rc, want = mico.pics_compress(ct, 65535, 4, 2)
    assert rc == 0 and mic.compress_parallel_strips(ct, w, h, 65535, 4) == want
    px, gw, gh = mic.decompress_parallel_strips(want)
    assert (gw, gh) == (w, h) and np.array_equal(np.asarray(px).reshape(h, w), ct)


@pytest.mark.parametrize("flavour", [1, 2, 4, 8, 108])
def test_fse_stage_is_the_same_stream_call_after_call(mic, mico, synth, gpu_ready, flavour):
    """Determinism of the 512-thread tANS encoder: its threads hand end states, fix-up records and the bits that reach into a
    neighbour's 64-bit unit to each other; a missed hand-off shows as a stream that changes from call to call (a development build
    did, for one-state streams, one call in two).  Several lengths, several calls each, every stream compared with the oracle's."""
    tok = mico.delta_rle_compress(synth.xr_like(cols=500, rows=180, depth=12, seed=17), 4095)
    for n in (tok.size, 65536, 60000, 40000, 20000):
        t = tok[:n].copy()
        rc, want = mico.fse_compress(t, flavour)
        assert rc == 0
        for _ in range(4):
            assert mic.fse_compress_u16(t, flavour) == want, n


def test_large_noisy_units_through_the_two_state_instance(mic, mico, synth, gpu_ready):
    """The 512-thread tANS encoder keeps the two-state walk in a kernel of its own; a unit whose two-state attempt gives up would be
    handed to the instance that holds the other walks and start its N -> ... -> 1 chain over there (multiframecompress.go:15-93).
    At tableLog <= 13 a token costs at most 13 bits, so "no gain" needs a header as large as the payload's slack: a scan of 12-bit
    noise frames from 64 x 48 to 256 x 192 found none (below ~150 x 110 the table build already fails, above it two states code) --
    the hand-over is there for completeness; this test pins the noisy large units on either side of it.  Whatever the oracle says
    -- a stream or an error code -- the GPU must say too."""
    noise = (synth.hash_u64(700 * 300, 5) & np.uint64(0xFFF)).astype(np.uint16).reshape(300, 700)
    smooth = synth.xr_like(cols=700, rows=300, depth=12, seed=4)
    mixed = np.where((np.arange(700)[None, :] // 50) % 2 == 0, noise, smooth).astype(np.uint16)
    small = (synth.hash_u64(140 * 100, 5) & np.uint64(0xFFF)).astype(np.uint16).reshape(100, 140)      # (the table build fails: -8)
    edge = (synth.hash_u64(160 * 120, 5) & np.uint64(0xFFF)).astype(np.uint16).reshape(120, 160)       # (the first size that codes)
    for img in (noise, mixed, small, edge):
        h, w = img.shape
        for ns in (2, 4, 8):
            rc, want = mico.compress_single_frame(img, 4095, ns)
            if rc == 0:
                got = mic.compress_single_frame(img, w, h, 4095, ns)
                assert got == want
                assert np.array_equal(mic.decompress_single_frame(got, w, h), img)
            else:
                with pytest.raises(mic.MicError) as e:
                    mic.compress_single_frame(img, w, h, 4095, ns)
                assert e.value.code == rc


def test_full_size_strips_are_the_same_streams_call_after_call_on_every_path(mic, mico, synth, gpu_ready):
    """VERDICT r3 / ADVICE r3: the determinism guard at the size the bench runs -- XR frames of 2577 x 2048 in eight strips (a strip is
    ~613 k tokens: ~1200 per thread of the 512-thread encoder, hand-offs at every thread and wave boundary) -- through the
    host-pointer batch entry point (session pool, sub-batch pipeline) and the single-image call, 2 / 4 / 8 states (the four- and
    eight-state walks live in the wide kernel instance), four calls each; every file equals the oracle's.  (No full-size input makes a
    two-state attempt fall back to one state -- see test_large_noisy_units_through_the_two_state_instance -- so that hand-over is
    covered at the sizes where it happens, there and in the small-frame parity cases.)"""
    w, h = 2577, 2048
    imgs = [synth.xr_like(cols=w, rows=h, depth=12, seed=900 + i, noise=synth.XR_NOISE_PUBLISHED_RATIO if i % 2 else 30.0) for i in range(3)]
    for ns in (2, 4, 8):
        want = []
        for im in imgs:
            rc, f = mico.pics_compress(im, 4095, 8, ns)
            assert rc == 0
            want.append(f)
        for call in range(4):
            res = mic.compress_parallel_strips_batch(imgs, 4095, 8, ns)
            for (st, blob), f in zip(res, want):
                assert st == 0 and blob.tobytes() == f, (ns, call)
        for call in range(2):
            assert mic.compress_parallel_strips(imgs[0], w, h, 4095, 8, ns) == want[0], (ns, call)

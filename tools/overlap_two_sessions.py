"""Experiment: does the step get shorter when the latency-bound tANS decode chain of one half of the batch runs
at the same time as the streaming kernels of the other half?  Two sessions (two HIP streams), each driven by its own host thread over
its own frames, against one session over all of them.  usage: python tools/overlap_two_sessions.py [frames] [steps]"""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
mic = entry.load_package()
synth = importlib.import_module("medical_image_codec_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 288
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W, H, S = 2577, 2048, 8
dev = torch.device("cuda:0")
d_all = synth.xr_like_batch_torch(B, cols=W, rows=H, depth=12, seed0=1, noise=synth.XR_NOISE_PUBLISHED_RATIO, device=dev)
sh = H // S


def make(d_px):
    n = d_px.shape[0]
    units = [(b * W * H + y0 * W, W, sh, 4095, 2) for b in range(n) for y0 in range(0, H, sh)]
    sess = mic.Session(len(units), W * sh)
    return sess, mic.Session.make_units(units), torch.empty_like(d_px)


ONLY = os.environ.get("ONLY", "")                  # ONLY=enc: the encode chain alone (tokeniser beside the other half's tANS encoder?)


def step(sess, cu, d_px, d_out):
    sess.encode_enqueue(d_px.data_ptr(), cu)
    d_blobs, offs, st, _ = sess.encode_finish()
    if ONLY == "enc":
        return
    sess.decode_enqueue(d_blobs, offs, cu, d_out.data_ptr())
    assert (sess.decode_finish() == 0).all()


def run(parts, label):
    ctx = [make(p) + (p,) for p in parts]
    for sess, cu, d_out, p in ctx:
        step(sess, cu, p, d_out)
    torch.cuda.synchronize()
    bar = threading.Barrier(len(ctx) + 1)

    def worker(c):
        sess, cu, d_out, p = c
        bar.wait()
        for _ in range(K):
            step(sess, cu, p, d_out)
        bar.wait()
    th = [threading.Thread(target=worker, args=(c,)) for c in ctx]
    for t in th:
        t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); el = time.perf_counter() - t0
    for t in th:
        t.join()
    for sess, cu, d_out, p in ctx:
        assert ONLY == "enc" or torch.equal(d_out, p)
        sess.close()
    raw = sum(p.numel() * 2 for p in parts) * K
    print(f"{label}: {el / K * 1e3:.2f} ms per round of {sum(p.shape[0] for p in parts)} frames, {raw / el / 1e9:.1f} GB/s", flush=True)


run([d_all], f"one session x {B} frames")
run([d_all[: B // 2], d_all[B // 2:]], f"two sessions x {B // 2} frames, concurrent")
run([d_all[: B // 3], d_all[B // 3: 2 * B // 3], d_all[2 * B // 3:]], f"three sessions x {B // 3} frames, concurrent")
d_b = d_all.clone()
run([d_all, d_b], f"two sessions x {B} frames, concurrent")

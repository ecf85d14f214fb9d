#!/usr/bin/env python3
"""bench.py -- PICS-8 encode+decode of XR-shaped 16-bit frames on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 it is launched
through torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): a batch of synthetic XR-like frames of the reference's
XR shape (cols 2577 x rows 2048, fseu16_test.go:32), each coded as PICS with 8 strips
(CompressParallelStrips(..., numStrips=8)).  One step = encode the whole batch, then decode
it again, with the frames already resident in HBM and the results left in HBM.  Strips are
independent, so each GPU codes its own batch (weak scaling, no data-path collective).

value        = raw u16 bytes of the batch (all ranks) / time of one encode+decode step.
roofline     = dominant kernel: algorithmic bytes (raw + compressed) per launch
               / its mean HIP-event duration on the session stream, against 8 TB/s HBM.
cpu_baseline = the reference's own C codec (ojph/mic_compress_c.c + mic_decompress_c.c, built in place into
               oracle/_ref/libmic_ref.so; kind "reference") coding the strips of the same frames on the host
               cores, one strip per thread (the mic_parallel.c model); the CPU oracle (kind "port") is timed the
               same way and reported beside it, and stands in when the reference build is not there.
"""
import argparse
import importlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def pics_strips(width, height, num_strips):
    """parallelstrips.go:59-72"""
    num_strips = max(1, min(num_strips, height))
    sh = (height + num_strips - 1) // num_strips
    actual = (height + sh - 1) // sh
    return [(i * sh, min(height, (i + 1) * sh)) for i in range(actual)]


def load_reference_codec():
    """the reference's C codec as built by `make -C oracle ref` (mic_compress_c.h:26-38, mic_decompress_c.h:24-50), or None"""
    import ctypes as C
    path = os.path.join(ROOT, "oracle", "_ref", "libmic_ref.so")
    if not os.path.exists(path):
        return None
    try:
        L = C.CDLL(path)
        L.mic_compress_two_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mic_decompress_two_state_simd.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
        return L
    except (OSError, AttributeError):
        return None


def cpu_baseline(mico, img, maxv, strips, budget_s=12.0, ref=None):
    """PICS-8 style encode+decode of one frame's strips, one strip per thread: with the reference C codec when ref is given
    (two-state encode, its SIMD two-state decode), else with the oracle."""
    import ctypes as C
    h, w = img.shape
    bounds = pics_strips(w, h, strips)
    cores = max(1, min(len(bounds), os.cpu_count() or 1))
    parts = [np.ascontiguousarray(img[a:b]) for a, b in bounds]

    if ref is not None:
        def enc(p):
            out = np.empty(p.size * 4 + 135168, dtype=np.uint8)
            n = C.c_size_t(0)
            rc = ref.mic_compress_two_state(p.ctypes.data, p.shape[1], p.shape[0], out.ctypes.data, out.size, C.byref(n))
            assert rc == 0
            return out[: n.value]

        def dec(args):
            blob, p = args
            px = np.empty_like(p)
            rc = ref.mic_decompress_two_state_simd(blob.ctypes.data, blob.size, px.ctypes.data, p.shape[1], p.shape[0])
            assert rc == 0 and np.array_equal(px, p)
            return 0
    else:
        def enc(p):
            rc, blob = mico.compress_single_frame(p, maxv, 2)
            assert rc == 0
            return blob

        def dec(args):
            blob, p = args
            rc, px = mico.decompress_single_frame(blob, p.shape[1], p.shape[0])
            assert rc == 0 and np.array_equal(px, p)
            return 0

    reps, t_enc, t_dec = 0, 0.0, 0.0
    t_start = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        while reps < 1 or (time.perf_counter() - t_start) < budget_s:
            t0 = time.perf_counter(); blobs = list(ex.map(enc, parts)); t1 = time.perf_counter()
            list(ex.map(dec, zip(blobs, parts))); t2 = time.perf_counter()
            t_enc += t1 - t0; t_dec += t2 - t1; reps += 1
    raw = img.nbytes * reps
    who = "reference C codec (ojph/mic_compress_c.c two-state, mic_decompress_two_state_simd)" if ref is not None else "oracle (C port of the Go path)"
    return {"value": raw / (t_enc + t_dec) / 1e9, "unit": "GB/s", "cores": cores, "kind": "reference" if ref is not None else "port",
            "encode_GBps": raw / t_enc / 1e9, "decode_GBps": raw / t_dec / 1e9,
            "sample": f"{reps} x encode+decode of the {len(parts)} strips of one {w}x{h} frame, {who}, one strip per thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images", type=int, default=int(os.environ.get("MIC_BENCH_IMAGES", "256")),
                    help="frames per GPU per step (256 x 8 strips = 8 tANS decode streams, 4 two-stream waves, on each of the 256 CUs)")
    ap.add_argument("--strips", type=int, default=8)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--cols", type=int, default=2577)
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    mic = entry.load_package()
    synth = importlib.import_module("medical_image_codec_amd.synth")
    rc = mic.lib().mic_hip_set_device(local)
    if rc:
        raise SystemExit(f"mic_hip_set_device({local}) rc={rc}: libmic_hip.so needs a gfx950 device (no CPU fallback)")

    W, H, B, S = args.cols, args.rows, args.images, args.strips
    maxv = (1 << args.depth) - 1
    # a few distinct frames, tiled over the batch (seed differs per rank)
    distinct = min(B, 4)
    base = [synth.xr_like(cols=W, rows=H, depth=args.depth, seed=1 + rank * 16 + i) for i in range(distinct)]
    host = np.stack([base[i % distinct] for i in range(B)])
    d_px = torch.from_numpy(host.view(np.int16)).to(dev)
    d_out = torch.empty_like(d_px)
    bounds = pics_strips(W, H, S)
    units = []
    for b in range(B):
        for (y0, y1) in bounds:
            units.append((b * W * H + y0 * W, W, y1 - y0, maxv, 2))
    n_units = len(units)
    max_px = max(u[1] * u[2] for u in units)
    sess = mic.Session(n_units, max_px)
    cunits = mic.Session.make_units(units)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(timing=False):
        sess.set_timing(timing)
        sess.encode_enqueue(d_px.data_ptr(), cunits)
        t_enc_k = sess.last_timings() if timing else None
        d_blobs, offs, st, ns = sess.encode_finish()
        assert (st == 0).all(), f"encode status {st[st != 0][:4]}"
        sess.decode_enqueue(d_blobs, offs, cunits, d_out.data_ptr())
        t_dec_k = sess.last_timings() if timing else None
        dst = sess.decode_finish()
        assert (dst == 0).all(), f"decode status {dst[dst != 0][:4]}"
        return offs, t_enc_k, t_dec_k

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        offs, _, _ = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # lossless check of the last step (outside the timed region)
    assert torch.equal(d_out, d_px), "round trip differs"
    comp_bytes = int(offs[-1])
    raw_bytes = host.nbytes
    ratio = raw_bytes / comp_bytes

    # per-kernel device times (HIP events on the session stream), separate instrumented steps
    kt = {}
    reps = 3
    for _ in range(reps):
        _, te, td = step(timing=True)
        for name, ms in (te + td):
            kt.setdefault(name, []).append(ms)
    kmean = {k: float(np.mean(v)) for k, v in kt.items()}
    dom = max(kmean, key=kmean.get)
    alg_bytes = raw_bytes + comp_bytes            # one direction: read raw + write compressed (or the reverse)
    achieved = alg_bytes / (kmean[dom] * 1e-3) / 1e9
    # HBM traffic of that kernel from the committed rocprofv3 PMC passes (profiles/, same command); only
    # quoted when the profile was taken on this very configuration
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        ent = tj.get(dom)
        if ent and ent.get("frames_per_gpu") == B and ent.get("width") == W and ent.get("height") == H and ent.get("depth") == args.depth:
            traffic = ent["hbm_bytes_per_launch"]
    except Exception:
        traffic = None

    ms_step = elapsed / args.steps * 1e3
    value = raw_bytes * world / (elapsed / args.steps) / 1e9
    enc_ms = sum(v for k, v in kmean.items() if k.startswith("k_enc") or k.startswith("k_scan"))
    dec_ms = sum(v for k, v in kmean.items() if k.startswith("k_dec"))

    out = {
        "metric": "PICS-8 encode+decode throughput over raw u16 bytes (XR-shaped frames), lossless",
        "value": round(value, 4), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u16", "data": "synthetic",
        "config": {"workload": f"PICS-{S} encode+decode, XR-like {W}x{H} {args.depth}-bit synthetic frames, {B} frames ({n_units} strips) per GPU per step",
                   "frames_per_gpu": B, "strips_per_frame": S, "width": W, "height": H, "max_value": maxv,
                   "fse": "2-state (CompressParallelStrips default)", "parallelism": f"{world} x independent batches"},
        "ratio": round(ratio, 4),
        "encode_GBps_kernels": round(raw_bytes / (enc_ms * 1e-3) / 1e9, 4) if enc_ms else None,
        "decode_GBps_kernels": round(raw_bytes / (dec_ms * 1e-3) / 1e9, 4) if dec_ms else None,
        "kernel_ms": {k: round(v, 4) for k, v in kmean.items()},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 4), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(kmean[dom], 4)},
    }
    if rank == 0 and not args.no_cpu and world == 1:
        from oracle import mico
        mico.lib()
        ref = load_reference_codec()
        port = cpu_baseline(mico, base[0], maxv, S, budget_s=8.0)
        if ref is not None:
            out["cpu_baseline"] = cpu_baseline(mico, base[0], maxv, S, budget_s=10.0, ref=ref)
            out["cpu_baseline"]["port"] = {k: port[k] for k in ("value", "encode_GBps", "decode_GBps", "cores")}
        else:
            out["cpu_baseline"] = port
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    sess.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

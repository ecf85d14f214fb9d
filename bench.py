#!/usr/bin/env python3
"""bench.py -- PICS-8 encode+decode of XR-shaped 16-bit frames on MI355X, plus the other BASELINE.json configurations.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 it is launched through
torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

Headline (BASELINE.json configs[1]): a batch of DISTINCT synthetic XR-like frames of the reference's XR shape (cols 2577 x
rows 2048, fseu16_test.go:32), noise tuned so that PICS-8 codes them at the reference's published XR ratio (~1.755), each coded
as PICS with 8 strips (CompressParallelStrips(..., numStrips=8)).  One step = encode the whole batch, then decode it again, with
the frames already resident in HBM and the results left in HBM.  Strips are independent, so each GPU codes its own batch (weak
scaling, no data-path collective); for N > 1 the assembly of rank 0's view of all streams (sizes all-gather + device-to-device
gather over RCCL) is timed separately under "container_assembly".

value        = raw u16 bytes of the batch (all ranks) / time of one encode+decode step.
roofline     = dominant kernel: algorithmic bytes (raw + compressed of one direction) per launch / its mean
               HIP-event duration on the session stream, against 8 TB/s HBM.
legs         = the same measurement for BASELINE configs 3 (WaveletV2 on CR-shaped frames), 4 (MIC2, 512 frames of 512 x 512)
               and 5 (MIC3, 32768 x 32768 RGB, 256 x 256 tiles), device-resident, each with its own roofline block (N = 1 only).
batch_sweep  = the headline at B in {1, 64, 288, 512} frames per launch, kernel time.
cpu_baseline = the reference's own C codec (ojph/mic_compress_c.c + mic_decompress_c.c, built in place into
               oracle/_ref/libmic_ref.so; kind "reference") coding the strips of one of the frames on the host cores, one
               strip per thread (the mic_parallel.c model: eight threads), and -- "all_cores" -- as many frames in flight as the
               host has cores / 8, one strip per thread on every core; "host" names the box (nproc, CPU model); the CPU oracle
               (kind "port") is timed beside it.
determinism  = two consecutive encode steps must produce byte-identical streams for every unit (asserted, outside the timed region).
end_to_end   = host buffers in and out through the C ABI's batch entry points; for N > 1 rank 0 drives all N devices from ONE
               process (mic_hip_set_devices), which is what a Go host would do: host-path GB/s over 1 / 2 / 4 / 8 PCIe links.
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
            "sample": f"{reps} x encode+decode of the {len(parts)} strips of one {w}x{h} frame, {who}, one strip per thread ({cores} threads)"}


def host_info():
    """core count and CPU model of the box the CPU figures come from"""
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return {"nproc": os.cpu_count() or 1, "usable": usable, "model": model}


def cpu_all_cores(img, strips, ref, budget_s=8.0):
    """Every usable core busy, the way a Go host would keep them: cores / strips frames in flight, one strip per thread (the model of
    ojph/mic_parallel.c:146-181 run from as many goroutines as there are frames).  Reference C codec only."""
    import ctypes as C
    h, w = img.shape
    bounds = pics_strips(w, h, strips)
    cores = host_info()["usable"]
    frames = max(1, cores // len(bounds))
    threads = min(cores, frames * len(bounds))
    imgs = [img] + [np.ascontiguousarray(np.roll(img, 17 * i, axis=1)) for i in range(1, frames)]     # (own memory per frame in flight)
    parts = [np.ascontiguousarray(im[a:b]) for im in imgs for a, b in bounds]

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
        assert rc == 0
        return 0

    reps, t_enc, t_dec = 0, 0.0, 0.0
    t_start = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        while reps < 1 or (time.perf_counter() - t_start) < budget_s:
            t0 = time.perf_counter(); blobs = list(ex.map(enc, parts)); t1 = time.perf_counter()
            list(ex.map(dec, zip(blobs, parts))); t2 = time.perf_counter()
            t_enc += t1 - t0; t_dec += t2 - t1; reps += 1
    raw = img.nbytes * frames * reps
    return {"value": raw / (t_enc + t_dec) / 1e9, "unit": "GB/s", "cores": threads, "frames_in_flight": frames,
            "encode_GBps": raw / t_enc / 1e9, "decode_GBps": raw / t_dec / 1e9,
            "sample": f"{reps} x encode+decode of {frames} frames x {len(bounds)} strips, reference C codec, one strip per thread on {threads} threads"}


def csrc_fingerprint():
    """sha256 over the kernel sources: a profile's counters describe the build they were taken on, and no other"""
    import hashlib
    d = os.path.join(ROOT, "medical-image-codec_amd", "csrc")
    hsh = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            hsh.update(name.encode()); hsh.update(open(os.path.join(d, name), "rb").read())
    return hsh.hexdigest()[:16]


def roofline_block(kmean, raw_bytes, comp_bytes, traffic=None):
    """dominant kernel of a leg against the HBM roofline: algorithmic bytes of ONE direction (raw + compressed) / its duration"""
    if not kmean:
        return None
    dom = max(kmean, key=kmean.get)
    alg = raw_bytes + comp_bytes
    ach = alg / (kmean[dom] * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dom, "achieved": round(ach, 4), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBPS, 6), "traffic": traffic, "algorithmic_bytes_per_launch": alg,
            "kernel_ms": round(kmean[dom], 4)}


def mean_timings(runs):
    kt = {}
    for t in runs:
        for name, ms in t:
            kt.setdefault(name, []).append(ms)
    return {k: float(np.mean(v)) for k, v in kt.items()}


def unit_codec_leg(mic, torch, d_px, units, steps, warmup, what):
    """encode + decode of a unit batch through a session; returns the leg's result dict"""
    n_units = len(units)
    max_px = max(u[1] * u[2] for u in units)
    sess = mic.Session(n_units, max_px)
    cunits = mic.Session.make_units(units)
    d_out = torch.empty_like(d_px)

    def step(timing=False):
        sess.set_timing(timing)
        sess.encode_enqueue(d_px.data_ptr(), cunits)
        te = sess.last_timings() if timing else []
        d_blobs, offs, st, _ = sess.encode_finish()
        assert (st == 0).all(), f"{what}: encode status {st[st != 0][:4]}"
        sess.decode_enqueue(d_blobs, offs, cunits, d_out.data_ptr())
        td = sess.last_timings() if timing else []
        dst = sess.decode_finish()
        assert (dst == 0).all(), f"{what}: decode status {dst[dst != 0][:4]}"
        return offs, te + td

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        offs, _ = step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    el = float(np.mean(times))
    assert torch.equal(d_out, d_px), f"{what}: round trip differs"
    kmean = mean_timings([step(True)[1] for _ in range(2)])
    sess.close()
    raw = d_px.numel() * 2
    comp = int(offs[-1])
    enc_ms = sum(v for k, v in kmean.items() if k.startswith("k_enc") or k.startswith("k_scan"))
    dec_ms = sum(v for k, v in kmean.items() if k.startswith("k_dec"))
    return {"value": round(raw / el / 1e9, 4), "unit": "GB/s", "ms_per_step": round(el * 1e3, 3), "steps": steps,
            "ms_min": round(min(times) * 1e3, 3), "ms_median": round(float(np.median(times)) * 1e3, 3), "ratio": round(raw / comp, 4),
            "raw_bytes": raw, "units": n_units,
            "encode_GBps_kernels": round(raw / (enc_ms * 1e-3) / 1e9, 4) if enc_ms else None,
            "decode_GBps_kernels": round(raw / (dec_ms * 1e-3) / 1e9, 4) if dec_ms else None,
            "kernel_ms": {k: round(v, 4) for k, v in kmean.items() if v >= 0.02},
            "roofline": roofline_block(kmean, raw, comp)}, kmean, offs


def leg_wavelet(mic, torch, synth, dev, steps, warmup, nframes=256):
    """BASELINE config 3: WaveletV2SIMDRLEFSECompressU16, 5 levels, CR shape rows 2140 x cols 1760, frames side by side"""
    rows, cols = 2140, 1760
    d_px = synth.xr_like_batch_torch(nframes, cols=cols, rows=rows, depth=12, seed0=2000, noise=5.0, device=dev)   # = synth.cr_like noise
    d_out = torch.empty_like(d_px)
    sess = mic.Session(nframes, 2 * rows * cols + 16)

    def step(timing=0):
        sess.set_timing(timing)
        d_s, offs, st, applied = sess.wavelet_v2_encode(d_px.data_ptr(), nframes, rows, cols, 5)
        te = sess.last_timings() if timing else []
        assert (st == 0).all() and applied == 5
        sess.set_timing(timing)
        dst = sess.wavelet_v2_decode(d_s, offs, nframes, rows, cols, applied, d_out.data_ptr())
        td = sess.last_timings() if timing else []
        assert (dst == 0).all()
        return offs, te + td

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        offs, _ = step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    el = float(np.mean(times))
    assert torch.equal(d_out, d_px), "wavelet round trip differs"
    kmean = mean_timings([step(2)[1] for _ in range(2)])
    sess.close()
    raw, comp = d_px.numel() * 2, int(offs[-1]) + 11 * nframes
    return {"workload": f"WaveletV2 (5 levels) encode+decode, {nframes} CR-like {cols}x{rows} 12-bit synthetic frames per launch, device-resident",
            "value": round(raw / el / 1e9, 4), "unit": "GB/s", "ms_per_step": round(el * 1e3, 3), "steps": steps,
            "ms_min": round(min(times) * 1e3, 3), "ms_median": round(float(np.median(times)) * 1e3, 3), "ratio": round(raw / comp, 4), "raw_bytes": raw,
            "kernel_ms": {k: round(v, 4) for k, v in kmean.items() if v >= 0.02}, "roofline": roofline_block(kmean, raw, comp)}


def leg_mic2(mic, torch, synth, dev, steps, warmup):
    """BASELINE config 4: CompressMultiFrame independent mode, 512 frames of 512 x 512 (multiframecompress.go:179-261)"""
    stack = synth.ct_stack(512, 512, 12, seed=3)
    d_px = torch.from_numpy(stack.view(np.int16)).to(dev)
    units = [(i * 512 * 512, 512, 512, 4095, 2) for i in range(512)]
    res, _, offs = unit_codec_leg(mic, torch, d_px, units, steps, warmup, "MIC2")
    # container assembly (WriteMIC2, multiframe.go:49-91): 20-byte header + 8 bytes per frame in front of the packed streams
    t0 = time.perf_counter()
    hdr = bytearray(20 + 8 * 512)
    hdr[0:4] = b"MIC2"; hdr[4:8] = (512).to_bytes(4, "little"); hdr[8:12] = (512).to_bytes(4, "little"); hdr[12:16] = (512).to_bytes(4, "little"); hdr[16] = 1
    for i in range(512):
        hdr[20 + 8 * i: 24 + 8 * i] = int(offs[i]).to_bytes(4, "little"); hdr[24 + 8 * i: 28 + 8 * i] = int(offs[i + 1] - offs[i]).to_bytes(4, "little")
    res["container_assembly_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    res["workload"] = "MIC2 independent mode encode+decode, synthetic CT-like stack 512 x 512 x 512 (12-bit), frames as units, device-resident"
    return res


def leg_wsi(mic, torch, synth, dev, steps, size=32768):
    """BASELINE config 5: CompressWSI, 8-bit RGB, 256 x 256 tiles, full pyramid (wsicompress.go:27-171); decode = every tile of every level"""
    slide = synth.wsi_slide(size, size, seed=4, workers=min(16, os.cpu_count() or 8))
    d_px = torch.from_numpy(slide).to(dev)
    del slide
    sess = mic.Session(1, 256 * 256)
    outs = None

    def step(timing=0):
        nonlocal outs
        sess.set_timing(timing)
        t0 = time.perf_counter()
        tiles, nbytes = sess.wsi_encode(d_px.data_ptr(), size, size)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        te = sess.last_timings() if timing else []
        lv = sess.wsi_levels()
        if outs is None:
            outs = [torch.empty((h, w, 3), dtype=torch.uint8, device=dev) for (w, h) in lv]
        sess.set_timing(timing)
        t2 = time.perf_counter()
        for k in range(len(lv)):
            sess.wsi_decode_level(k, outs[k].data_ptr(), outs[k].numel())
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        td = sess.last_timings() if timing else []
        return tiles, nbytes, lv, t1 - t0, t3 - t2, te + td

    step()
    enc_t, dec_t = [], []
    for _ in range(steps):
        tiles, nbytes, lv, a, b, _ = step()
        enc_t.append(a); dec_t.append(b)
    assert torch.equal(outs[0], d_px), "WSI level-0 round trip differs"
    kmean = mean_timings([step(2)[5]])
    fbuf = mic.host_alloc(nbytes + 64)                                   # the caller's file buffer (pinned: WriteMIC3 is one DMA + 0.4 MB of index)
    t_writes = []
    for _ in range(3):
        t0 = time.perf_counter()
        flen = sess.wsi_write(fbuf)
        t_writes.append(time.perf_counter() - t0)
    t_write = min(t_writes)
    assert flen == nbytes
    pbuf = np.empty(nbytes + 64, dtype=np.uint8)                          # the same into ordinary memory (staged by the transfer threads)
    sess.wsi_write(pbuf)
    t0 = time.perf_counter(); sess.wsi_write(pbuf); t_write_pageable = time.perf_counter() - t0
    assert np.array_equal(pbuf[:nbytes], fbuf[:nbytes])
    mic.host_free(fbuf); del pbuf
    sess.close()
    raw_l0 = size * size * 3
    tile_rgb_bytes = sum(((w + 255) // 256) * ((h + 255) // 256) for w, h in lv) * 256 * 256 * 3       # RGB bytes of every tile incl. padding, all levels
    el = float(np.mean(enc_t) + np.mean(dec_t))
    tot_t = [a + b for a, b in zip(enc_t, dec_t)]
    # 3 N (1 + 1/r) with N = tile pixels including padding (all levels), r = that / compressed bytes
    return {"workload": f"MIC3 encode (pyramid + tiles + planes) and decode (every tile of every level), synthetic H&E-like {size}x{size} RGB slide, "
                        f"256x256 tiles, {len(lv)} levels, {tiles} tiles, device-resident",
            "value": round(raw_l0 / el / 1e9, 4), "unit": "GB/s (level-0 RGB bytes / encode+decode time)", "ms_per_step": round(el * 1e3, 3),
            "steps": steps, "ms_min": round(min(tot_t) * 1e3, 3), "ms_median": round(float(np.median(tot_t)) * 1e3, 3),
            "encode_ms": round(float(np.mean(enc_t)) * 1e3, 3), "decode_ms": round(float(np.mean(dec_t)) * 1e3, 3),
            "ratio": round(raw_l0 / nbytes, 4), "raw_bytes": raw_l0, "compressed_bytes": nbytes,
            "container_assembly_ms": round(t_write * 1e3, 3), "container_assembly_pageable_ms": round(t_write_pageable * 1e3, 3),
            "kernel_ms": {k: round(v, 4) for k, v in kmean.items() if v >= 0.05},
            "roofline": roofline_block(kmean, tile_rgb_bytes, nbytes)}


def legs_multi_gpu(mic, torch, synth, par, dist, dev, rank, world, steps=5):
    """BASELINE configs 4 and 5 as BASELINE.json words them ("sharded across 8 MI355X"), for N > 1: every rank codes its shard
    (frames of the 512^3 stack: multiframecompress.go:201-203; a band of the 32768^2 slide: wsicompress.go:126-145), rank 0
    assembles the container.  `coding` = the slowest rank's own encode (+ decode) of its shard; `container_assembly` = everything
    the distributed call adds to it: sizes all-gather, device-to-device gather of the streams over RCCL, the top pyramid levels and the
    header on rank 0.  min / median over `steps` runs."""
    def barrier():
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()

    def tmax(v):
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def stats(xs):
        return {"min_ms": round(min(xs) * 1e3, 3), "median_ms": round(float(np.median(xs)) * 1e3, 3)}

    out = {}
    # ---- config 4: MIC2 independent mode, 512 frames of 512 x 512, frames sharded
    n, sz = 512, 512
    lo, hi = par.shard_range(n, world, rank)
    stack = synth.ct_stack(n, sz, 12, seed=3)[lo:hi]
    d_px = torch.from_numpy(np.ascontiguousarray(stack).view(np.int16)).to(dev)
    units = [((i - lo) * sz * sz if lo <= i < hi else 0, sz, sz, 4095, 2) for i in range(n)]
    sess = mic.Session(max(hi - lo, 1), sz * sz, device=dev.index)
    enc, dec = par.session_codec(mic, sess, d_px, units)
    t_code, t_all, t_dcode, t_dall = [], [], [], []
    mic2 = None
    for it in range(steps + 1):
        barrier(); t0 = time.perf_counter(); blobs, sizes = enc(lo, hi); torch.cuda.synchronize(); a = tmax(time.perf_counter() - t0)
        barrier(); t0 = time.perf_counter(); mic2 = par.dist_compress_multi_frame(enc, sz, sz, n); barrier(); b = tmax(time.perf_counter() - t0)
        loffs = np.concatenate([[0], np.cumsum(sizes.cpu().numpy())]).astype(np.int64)
        barrier(); t0 = time.perf_counter(); px = dec(lo, hi, blobs, loffs, sz, sz); torch.cuda.synchronize(); c = tmax(time.perf_counter() - t0)
        barrier(); t0 = time.perf_counter(); _, _, px2 = par.dist_decompress_multi_frame(dec, mic2, device=dev); barrier(); d = tmax(time.perf_counter() - t0)
        assert torch.equal(px, d_px.view(px.shape)) and torch.equal(px2, d_px.view(px2.shape)), "MIC2 shard round trip differs"
        if it:
            t_code.append(a); t_all.append(b); t_dcode.append(c); t_dall.append(d)
    sess.close()
    raw = n * sz * sz * 2
    out["config4_mic2_512cubed"] = {
        "workload": f"MIC2 independent mode, 512 x 512 x 512 CT-like stack, {hi - lo} frames per rank on {world} ranks",
        "value": round(raw / (min(t_code) + min(t_dcode)) / 1e9, 4), "unit": "GB/s (coding of the shards, slowest rank)",
        "encode_coding": stats(t_code), "encode_with_assembly": stats(t_all), "decode_coding": stats(t_dcode), "decode_with_scatter": stats(t_dall),
        "container_assembly_ms": round((min(t_all) - min(t_code)) * 1e3, 3),
        "ratio": round(raw / len(mic2), 4) if mic2 else None,
        "note": "512 frames are 512 entropy chains: 64 per GPU on 8 GPUs is a quarter chain per CU -- latency-bound, the configuration is too small for the node"}
    del d_px
    # ---- config 5: MIC3, 32768 x 32768 RGB, bands of tile rows
    size = 32768
    levels = par.wsi_levels(size, size, 256, 256, 0)
    K, bands = par.wsi_band_plan(size, 256, len(levels), world)
    y0, y1 = bands[rank]
    band = torch.from_numpy(synth.wsi_slide_band(size, size, y0, y1, seed=4)).to(dev)
    sess = mic.Session(1, 256 * 256, device=dev.index)
    enc_slide = par.session_wsi_codec(mic, sess)
    t_code, t_all = [], []
    f = None
    fbuf = mic.host_alloc(size * size * 3 // 2 + (1 << 20)) if rank == 0 else None      # rank 0's file buffer (pinned)
    for it in range(max(2, steps // 2) + 1):
        barrier(); t0 = time.perf_counter()
        if y1 > y0:
            enc_slide(band, min(K + 1, len(levels)))
        torch.cuda.synchronize(); a = tmax(time.perf_counter() - t0)
        barrier(); t0 = time.perf_counter(); f = par.dist_compress_wsi(enc_slide, band, size, size, out=fbuf); barrier(); b = tmax(time.perf_counter() - t0)
        if it:
            t_code.append(a); t_all.append(b)
    sess.close()
    if fbuf is not None:
        mic.host_free(fbuf)
    out["config5_mic3_wsi_32768"] = {
        "workload": f"MIC3 encode of a {size}x{size} RGB slide in bands of {256 << K} rows (levels 0..{K} per band, the rest on rank 0), {world} ranks",
        "value": round(size * size * 3 / min(t_code) / 1e9, 4), "unit": "GB/s of level-0 RGB (coding of the bands, slowest rank)",
        "encode_coding": stats(t_code), "encode_with_assembly": stats(t_all),
        "container_assembly_ms": round((min(t_all) - min(t_code)) * 1e3, 3),
        "compressed_bytes": int(f) if f else None}
    return out


def leg_end_to_end(mic, torch, d_px, W, H, S, maxv, dev, devices=None):
    """SURVEY.md §8(d) "Timing", the second figure: host buffers in, host buffers out, through the C ABI's batch entry points
    (mic_hip_pics_compress_batch / _decompress_batch -- what a cgo caller uses), for ordinary (pageable) numpy buffers and for
    pinned ones (mic_hip_host_alloc), at B = all frames and B = 1; beside it the box's pinned H2D / D2H copy rates, and the
    fraction of the PCIe floor (bytes in / H2D rate + bytes out / D2H rate) the call reaches."""
    B0 = d_px.shape[0]
    frame_bytes = W * H * 2
    host0 = d_px.cpu().numpy().view(np.uint16).reshape(B0, H, W)
    # several devices (ONE process, mic_hip_set_devices): the batch is B0 frames per device -- the same frames for every device, which
    # costs the host no memory and the codec nothing -- so that every link carries what the single link carries at N = 1
    ndev = len(devices) if devices else 1
    B = B0 * ndev
    if devices:
        mic.set_devices(devices)

    # the link: one 1 GiB pinned buffer each way
    n = 1 << 30
    pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    dv = torch.empty(n, dtype=torch.uint8, device=dev)
    rates = {}
    for name, fn in (("h2d", lambda: dv.copy_(pin, non_blocking=True)), ("d2h", lambda: pin.copy_(dv, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        rates[name] = 3 * n / (time.perf_counter() - t0) / 1e9
    del pin, dv

    def run(kind):
        if kind == "pinned":
            src = mic.host_alloc(B0 * frame_bytes, np.uint16).reshape(B0, H, W)
            src[...] = host0
            cbuf = mic.host_alloc(B * frame_bytes)
            back = mic.host_alloc(B * frame_bytes, np.uint16).reshape(B, H, W)
        else:
            src = host0
            cbuf = np.empty(B * frame_bytes, dtype=np.uint8)
            back = np.empty((B, H, W), dtype=np.uint16)
        imgs = [src[i % B0] for i in range(B)]
        outs = [cbuf[i * frame_bytes:(i + 1) * frame_bytes] for i in range(B)]
        pxo = [back[i].reshape(-1) for i in range(B)]
        best = None
        for it in range(3):                                     # the first pass pays for the session's workspace
            t0 = time.perf_counter()
            res = mic.compress_parallel_strips_batch(imgs, maxv, S, 2, outs=outs)
            t1 = time.perf_counter()
            assert all(st == 0 for st, _ in res)
            files = [b for _, b in res]
            t2 = time.perf_counter()
            dec = mic.decompress_parallel_strips_batch(files, [(W, H)] * B, outs=pxo)
            t3 = time.perf_counter()
            assert all(st == 0 for st, _ in dec)
            if it and (best is None or (t1 - t0) + (t3 - t2) < best[0] + best[1]):
                best = (t1 - t0, t3 - t2)
        for k in range(ndev):
            assert np.array_equal(back[k * B0:(k + 1) * B0], host0), "end-to-end round trip differs"
        comp = sum(len(f) for f in files)
        raw = B * frame_bytes
        floor_enc = raw / rates["h2d"] / 1e9 + comp / rates["d2h"] / 1e9
        floor_dec = comp / rates["h2d"] / 1e9 + raw / rates["d2h"] / 1e9
        # B = 1: one frame, one call each way
        t0 = time.perf_counter(); (st, one), = mic.compress_parallel_strips_batch(imgs[:1], maxv, S, 2, outs=outs[:1]); t1 = time.perf_counter()
        (st2, _), = mic.decompress_parallel_strips_batch([one], [(W, H)], outs=pxo[:1]); t2 = time.perf_counter()
        r = {"encode_GBps": round(raw / best[0] / 1e9, 3), "decode_GBps": round(raw / best[1] / 1e9, 3),
             "encode_decode_GBps": round(raw / (best[0] + best[1]) / 1e9, 3),
             "encode_ms": round(best[0] * 1e3, 2), "decode_ms": round(best[1] * 1e3, 2),
             "pcie_floor_ms": {"encode": round(floor_enc * 1e3, 2), "decode": round(floor_dec * 1e3, 2)},
             "fraction_of_pcie_floor": {"encode": round(floor_enc / best[0], 3), "decode": round(floor_dec / best[1], 3)},
             # (the floor above moves one way at a time; the link is full duplex, so a pipelined call can beat it.  Both ways at once:)
             "pcie_duplex_floor_ms": {"encode": round(max(raw / rates["h2d"], comp / rates["d2h"]) / 1e6, 2),
                                      "decode": round(max(comp / rates["h2d"], raw / rates["d2h"]) / 1e6, 2)},
             "b1_encode_ms": round((t1 - t0) * 1e3, 2), "b1_decode_ms": round((t2 - t1) * 1e3, 2)}
        if kind == "pinned":
            for a in (src, cbuf, back):
                mic.host_free(a.reshape(-1).view(np.uint8) if a.dtype != np.uint8 else a)
        return r

    out = {"what": f"mic_hip_pics_compress_batch + mic_hip_pics_decompress_batch, {B} frames ({B * S} strips) per call, host buffers in and out",
           "devices": list(devices) if devices else [dev.index if dev.index is not None else 0], "process": "one",
           "pcie_pinned_GBps": {k: round(v, 2) for k, v in rates.items()},
           "pcie_note": "link rates and floors are ONE device's; with N devices the floor of the call is that of a 1 / N share",
           "pageable": run("pageable")}
    if ndev == 1:
        out["pinned"] = run("pinned")
    if devices:
        mic.set_devices([devices[0]])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images", type=int, default=int(os.environ.get("MIC_BENCH_IMAGES", "288")),
                    help="frames per GPU per step (288 x 8 strips = 9 tANS decode streams on each of the 256 CUs)")
    ap.add_argument("--strips", type=int, default=8)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--cols", type=int, default=2577)
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--noise", type=float, default=None, help="xr_like noise (default: the level that codes at the published XR ratio 1.755)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip configs 3-5 and the batch sweep")
    ap.add_argument("--legs-only", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-buffer (PCIe-inclusive) figure")
    args = ap.parse_args()

    # stdout carries ONE line, the result.  Native libraries write there too (RCCL prints a version banner when a process group comes
    # up): while the bench runs, file descriptor 1 points at stderr; it is put back for the line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

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
    par = importlib.import_module("medical_image_codec_amd.parallel")
    rc = mic.lib().mic_hip_set_device(local)
    if rc:
        raise SystemExit(f"mic_hip_set_device({local}) rc={rc}: libmic_hip.so needs a gfx950 device (no CPU fallback)")

    W, H, B, S = args.cols, args.rows, args.images, args.strips
    maxv = (1 << args.depth) - 1
    noise = args.noise if args.noise is not None else synth.XR_NOISE_PUBLISHED_RATIO
    seed0 = 1 + rank * 100003
    d_px = synth.xr_like_batch_torch(B, cols=W, rows=H, depth=args.depth, seed0=seed0, noise=noise, device=dev)   # B distinct frames
    d_out = torch.empty_like(d_px)
    bounds = pics_strips(W, H, S)
    units = [(b * W * H + y0 * W, W, y1 - y0, maxv, 2) for b in range(B) for (y0, y1) in bounds]
    n_units = len(units)
    max_px = max(u[1] * u[2] for u in units)
    sess = mic.Session(n_units, max_px, device=local)
    cunits = mic.Session.make_units(units)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(timing=False):
        sess.set_timing(timing)
        sess.encode_enqueue(d_px.data_ptr(), cunits)
        t_enc_k = sess.last_timings() if timing else []
        d_blobs, offs, st, ns = sess.encode_finish()
        assert (st == 0).all(), f"encode status {st[st != 0][:4]}"
        sess.decode_enqueue(d_blobs, offs, cunits, d_out.data_ptr())
        t_dec_k = sess.last_timings() if timing else []
        dst = sess.decode_finish()
        assert (dst == 0).all(), f"decode status {dst[dst != 0][:4]}"
        return d_blobs, offs, t_enc_k + t_dec_k

    out = {}
    if not args.legs_only:
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            d_blobs, offs, _ = step()
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        assert torch.equal(d_out, d_px), "round trip differs"             # lossless check of the last step (outside the timed region)
        comp_bytes = int(offs[-1])
        # consecutive steps, byte for byte: a lossless codec's stream is a pure function of its input (fse2state.go:122-199,
        # fsecompressu16.go:81-187) -- every unit's bytes and every offset of the next steps equal this one's (outside the timed region)
        first = torch.empty(comp_bytes, dtype=torch.uint8, device=dev)
        mic.device_copy(first.data_ptr(), d_blobs, comp_bytes)
        offs_first = np.array(offs, copy=True)
        same_all = True
        for _ in range(2):
            d_blobs2, offs2, _ = step()
            again = torch.empty(int(offs2[-1]), dtype=torch.uint8, device=dev)
            mic.device_copy(again.data_ptr(), d_blobs2, int(offs2[-1]))
            same_all = same_all and np.array_equal(offs2, offs_first) and bool(torch.equal(again, first))
            del again
        assert same_all, "the encoder produced different bytes for the same input in consecutive steps"
        determinism = {"steps_compared": 3, "units": n_units, "bytes": comp_bytes, "identical": True}
        del first
        raw_bytes = d_px.numel() * 2
        kmean = mean_timings([step(True)[2] for _ in range(3)])           # per-kernel device times, separate instrumented steps
        dom = max(kmean, key=kmean.get)
        # HBM bytes of the dominant kernel: the committed PMC passes of this same command (tools/profile_bench.sh) -- used only when
        # they were taken on THIS build of the kernels (the profile records a hash of csrc/) and on this workload
        traffic, traffic_source = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r04_traffic.json")))
            meta = tj.get("_meta", {})
            ent = tj.get(dom)
            if meta.get("csrc_sha16") != csrc_fingerprint():
                traffic_source = {"file": "profiles/r04_traffic.json", "used": False, "why": "csrc/ changed since the profile was taken"}
            elif ent and ent.get("frames_per_gpu") == B and ent.get("width") == W and ent.get("height") == H and ent.get("depth") == args.depth:
                traffic = ent["hbm_bytes_per_launch"]
                traffic_source = {"file": "profiles/r04_traffic.json", "used": True, "csrc_sha16": meta.get("csrc_sha16")}
        except Exception:
            traffic = None
        ms_step = elapsed / args.steps * 1e3
        value = raw_bytes * world / (elapsed / args.steps) / 1e9
        enc_ms = sum(v for k, v in kmean.items() if k.startswith("k_enc") or k.startswith("k_scan"))
        dec_ms = sum(v for k, v in kmean.items() if k.startswith("k_dec"))

        # N > 1: rank 0's view of every rank's streams (what a PICS-per-frame / MIC2 writer on rank 0 needs): sizes all-gather and
        # a device-to-device gather of the packed blobs over RCCL, timed apart from the step
        assembly = None
        if dist is not None:
            mine = torch.empty(comp_bytes, dtype=torch.uint8, device=dev)
            mic.device_copy(mine.data_ptr(), d_blobs, comp_bytes)
            sizes = torch.from_numpy(np.diff(offs.astype(np.int64))).to(dev)
            barrier()
            t0 = time.perf_counter()
            allb, alloffs = par.gather_unit_blobs(mine, sizes, n_units * world, dst=0)
            barrier()
            assembly = {"ms": round((time.perf_counter() - t0) * 1e3, 3),
                        "bytes_on_rank0": int(alloffs[-1]) if rank == 0 else None,
                        "what": "all_gather of per-unit sizes + send/recv of the packed device blobs to rank 0 (RCCL over xGMI)"}

        out = {
            "metric": "PICS-8 encode+decode throughput over raw u16 bytes (XR-shaped frames), lossless",
            "value": round(value, 4), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": f"PICS-{S} encode+decode, {B} distinct XR-like {W}x{H} {args.depth}-bit synthetic frames ({n_units} strips) per GPU per step, "
                                   f"noise tuned to the published XR PICS-8 ratio",
                       "frames_per_gpu": B, "strips_per_frame": S, "width": W, "height": H, "max_value": maxv, "noise": noise,
                       "fse": "2-state (CompressParallelStrips default)", "parallelism": f"{world} x independent batches"},
            "ratio": round(raw_bytes / comp_bytes, 4),
            "encode_GBps_kernels": round(raw_bytes / (enc_ms * 1e-3) / 1e9, 4) if enc_ms else None,
            "decode_GBps_kernels": round(raw_bytes / (dec_ms * 1e-3) / 1e9, 4) if dec_ms else None,
            "kernel_ms": {k: round(v, 4) for k, v in kmean.items() if v >= 0.02},
            "roofline": roofline_block(kmean, raw_bytes, comp_bytes, traffic),
            "traffic_source": traffic_source,
            "determinism": determinism,
            "container_assembly": assembly,
            "workspace": {"session_bytes": sess.workspace_bytes()[0], "tier2": sess.workspace_bytes()[1],
                          "x_input": round(sess.workspace_bytes()[0] / raw_bytes, 2)},
        }
    sess.close()

    if world == 1 and os.environ.get("MIC_BENCH_REHEARSE_MULTI"):       # the N > 1 legs on one rank (nccl, world size 1): a rehearsal of the code path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        out["legs_multi_gpu_rehearsal"] = legs_multi_gpu(mic, torch, synth, par, dist, dev, 0, 1, steps=2)
        dist.destroy_process_group()
        dist = None
    if world > 1 and not args.no_legs:
        try:
            out["legs_multi_gpu"] = legs_multi_gpu(mic, torch, synth, par, dist, dev, rank, world)
        except Exception as e:  # noqa: BLE001 -- the headline line must come out whatever a leg does
            out["legs_multi_gpu"] = {"error": repr(e)}
        torch.cuda.empty_cache()

    if world == 1 and not args.no_e2e and not args.legs_only:
        try:
            out["end_to_end"] = leg_end_to_end(mic, torch, d_px, W, H, S, maxv, dev)
        except Exception as e:  # noqa: BLE001 -- the headline line must come out whatever a leg does
            out["end_to_end"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    if world > 1 and not args.no_e2e and not args.legs_only:
        # ONE process, all N devices (what a Go host does): rank 0 drives every GPU of the job through mic_hip_set_devices while the
        # other ranks wait at the barrier below with their sessions closed
        torch.cuda.empty_cache()
        dist.barrier()
        if rank == 0:
            try:
                out["end_to_end"] = leg_end_to_end(mic, torch, d_px, W, H, S, maxv, dev, devices=list(range(world)))
            except Exception as e:  # noqa: BLE001
                out["end_to_end"] = {"error": repr(e)}
        dist.barrier()
        torch.cuda.empty_cache()

    if world == 1 and not args.no_legs:
        # B in {1, 64, 288, 512}: kernel time of one encode+decode per batch size (report B = 1 honestly)
        sweep = []
        for b in (1, 4, 16, 64, 288, 512):
            if b > B:
                dsw = synth.xr_like_batch_torch(b, cols=W, rows=H, depth=args.depth, seed0=seed0, noise=noise, device=dev)
            else:
                dsw = d_px[:b].contiguous()
            us = [(i * W * H + y0 * W, W, y1 - y0, maxv, 2) for i in range(b) for (y0, y1) in bounds]
            r, km, _ = unit_codec_leg(mic, torch, dsw, us, 10 if b <= 64 else 5, 1, f"sweep B={b}")
            enc_k = sum(v for k, v in km.items() if k.startswith("k_enc") or k.startswith("k_scan"))
            dec_k = sum(v for k, v in km.items() if k.startswith("k_dec"))
            sweep.append({"frames": b, "strips": len(us), "GBps": r["value"], "ms_per_step": r["ms_per_step"], "ms_min": r["ms_min"], "ms_median": r["ms_median"],
                          "encode_kernels_ms": round(enc_k, 3), "decode_kernels_ms": round(dec_k, 3),
                          "encode_GBps_kernels": r["encode_GBps_kernels"], "decode_GBps_kernels": r["decode_GBps_kernels"]})
            del dsw
        out["batch_sweep"] = sweep
        # config 2 read literally ("encode+decode of XR_2577_2048_image.bin": ONE image) is eight serial entropy chains: say so up front
        out["b1"] = {"encode_kernels_ms": sweep[0]["encode_kernels_ms"], "decode_kernels_ms": sweep[0]["decode_kernels_ms"],
                     "ms_per_step": sweep[0]["ms_per_step"], "GBps": sweep[0]["GBps"],
                     "note": "one frame = 8 strips = 8 serial tANS chains of ~330k rounds: latency, whatever the chip's width"}
        del d_out
        torch.cuda.empty_cache()
        legs = {}

        def leg(name, fn):                                       # (the headline line must come out whatever a leg does)
            try:
                legs[name] = fn()
            except Exception as e:  # noqa: BLE001
                legs[name] = {"error": repr(e)}
            torch.cuda.empty_cache()
        leg("config3_wavelet_v2_cr", lambda: leg_wavelet(mic, torch, synth, dev, 10, 1))
        leg("config4_mic2_512cubed", lambda: leg_mic2(mic, torch, synth, dev, 10, 2))
        del d_px
        torch.cuda.empty_cache()
        leg("config5_mic3_wsi_32768", lambda: leg_wsi(mic, torch, synth, dev, 10))
        out["legs"] = legs
        if "roofline" in out:
            out["roofline_fracs"] = {"config2_pics8_xr": out["roofline"]["frac"], **{k: (v.get("roofline") or {}).get("frac") for k, v in legs.items()}}

    if rank == 0 and not args.no_cpu and not args.legs_only:
        from oracle import mico
        mico.lib()
        ref = load_reference_codec()
        frame0 = synth.xr_like(cols=W, rows=H, depth=args.depth, seed=seed0, noise=noise)      # == frame 0 of the device batch
        port = cpu_baseline(mico, frame0, maxv, S, budget_s=8.0)
        if ref is not None:
            out["cpu_baseline"] = cpu_baseline(mico, frame0, maxv, S, budget_s=8.0, ref=ref)
            out["cpu_baseline"]["port"] = {k: port[k] for k in ("value", "encode_GBps", "decode_GBps", "cores")}
            try:
                out["cpu_baseline"]["all_cores"] = cpu_all_cores(frame0, S, ref, budget_s=8.0)
            except Exception as e:  # noqa: BLE001 -- a box that refuses that many threads still gets its eight-thread figure
                out["cpu_baseline"]["all_cores"] = None
                out["cpu_baseline"]["all_cores_error"] = repr(e)
        else:
            out["cpu_baseline"] = port
        out["cpu_baseline"]["host"] = host_info()
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0 and out.get("cpu_baseline") and out.get("batch_sweep"):
        # smallest batch at which the device (frames resident in HBM) outruns this box's host cores on the same frames
        sw = out["batch_sweep"]

        def crossover(cpu):
            for a, b in zip(sw, sw[1:]):
                if a["GBps"] < cpu <= b["GBps"]:
                    f = (cpu - a["GBps"]) / (b["GBps"] - a["GBps"])
                    return int(np.ceil(a["frames"] + f * (b["frames"] - a["frames"])))
            return sw[0]["frames"] if sw[0]["GBps"] >= cpu else None
        ac = out["cpu_baseline"].get("all_cores")
        out["crossover_frames"] = crossover(ac["value"] if ac else out["cpu_baseline"]["value"])       # vs every core of the host
        out["crossover_frames_8_threads"] = crossover(out["cpu_baseline"]["value"])                    # vs one frame's eight strips
    if dist is not None:
        dist.barrier()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    os.dup2(2, 1)                                                        # (whatever the teardown prints is not the result)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

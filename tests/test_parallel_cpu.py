"""World-size-2 and -4 gloo tests (CPU) of the multi-GPU plumbing: static unit sharding, size all-gather, point-to-point blob gather /
scatter, MIC2 and PICS assembly, sharded decode.  The codec is a stand-in here (the oracle, on CPU tensors), exactly as the
mic_hip session is injected on a GPU node (parallel.session_codec; tests/test_gpu_parity.py runs that on one rank with `nccl`);
the assembled files must equal the single-process oracle's."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, load_package


def _oracle_codec(mico, stack, maxv, units=None):
    """EncodeUnits / DecodeUnits over frames (units = None) or over explicit (frame, y0, y1) strips, on CPU tensors"""
    import torch

    def encode(lo, hi):
        blobs = []
        for i in range(lo, hi):
            px = stack[i] if units is None else stack[units[i][0], units[i][1]:units[i][2]]
            rc, blob = mico.compress_single_frame(np.ascontiguousarray(px), maxv, 2)
            assert rc == 0
            blobs.append(blob)
        flat = np.frombuffer(b"".join(blobs), dtype=np.uint8).copy()
        return torch.from_numpy(flat), torch.tensor([len(b) for b in blobs], dtype=torch.int64)

    def decode(lo, hi, blobs, offs, w, h):
        host = blobs.numpy().tobytes()
        out = []
        for k in range(hi - lo):
            rc, px = mico.decompress_single_frame(host[int(offs[k]):int(offs[k + 1])], w, h)
            assert rc == 0
            out.append(px)
        return np.stack(out) if out else np.zeros((0, h, w), np.uint16)

    return encode, decode


def _worker(rank, world, port, stack, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    load_package()
    import importlib
    par = importlib.import_module("medical_image_codec_amd.parallel")
    from oracle import mico
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, h, w = stack.shape
        enc, dec = _oracle_codec(mico, stack, 4095)
        mic2 = par.dist_compress_multi_frame(enc, w, h, n)                       # encode: shard, gather sizes + blobs, assemble
        lo, hi, px = par.dist_decompress_multi_frame(dec, mic2)                   # decode: scatter the streams, decode the shard
        assert (lo, hi) == par.shard_range(n, world, rank) and np.array_equal(px, stack[lo:hi])
        ns = 3
        sh = (h + ns - 1) // ns
        strips = [(f, y0, min(h, y0 + sh)) for f in range(n) for y0 in range(0, h, sh)]
        enc_s, _ = _oracle_codec(mico, stack, 4095, strips)
        pics = par.dist_compress_pics_batch(enc_s, w, h, ns, n)                  # PICS batches: frames sharded, strips as units
        # MIC3: the slide in bands of tile rows (levels 0..2 per band), the top of the pyramid (level 3) on rank 0
        import torch
        slide = synth_slide()
        H, W = slide.shape[:2]
        lv = par.wsi_levels(W, H, 256, 256, 0)
        K, bands = par.wsi_band_plan(H, 256, len(lv), world)
        assert len(lv) == 4
        if world == 2:
            assert K == 2 and bands == [(0, 1024), (1024, 1300)]
        if world == 4:                                                           # (K changes with the world size: bands of one tile row)
            assert K == 0 and bands == [(0, 256), (256, 768), (768, 1024), (1024, 1300)]

        def encode_file(img, levels):
            rc, f = mico.wsi_compress(np.ascontiguousarray(img.numpy()), 256, 256, levels)
            assert rc == 0
            return f
        y0, y1 = bands[rank]
        mic3 = par.dist_compress_wsi(par.slide_codec_from_bytes(encode_file), torch.from_numpy(slide[y0:y1].copy()), W, H)
        if rank == 0:
            q.put((mic2, pics, mic3))
    finally:
        dist.destroy_process_group()


def synth_slide():
    """520 x 1300 RGB slide (ragged right and bottom edges), tiles of 256: four levels, bands of 1024 rows for two ranks"""
    import importlib
    return importlib.import_module("medical_image_codec_amd.synth").wsi_slide(520, 1300, seed=11, workers=1)


def test_shard_range_is_a_partition(mic):
    import importlib
    par = importlib.import_module("medical_image_codec_amd.parallel")
    for n in (1, 7, 8, 64, 512, 21845):
        for world in (1, 2, 4, 8):
            edges = [par.shard_range(n, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in edges) - min(b - a for a, b in edges) <= 1


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_mic2_pics_and_mic3_assembly_match_single_process(mico, synth, world):
    stack = np.stack([synth.xr_like(cols=128, rows=128, depth=12, seed=80 + i) for i in range(5)])   # (every strip codes: no constant one)
    rc, want = mico.mic2_compress(stack, 4095, False)
    assert rc == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, stack, q)) for r in range(world)]
    for p in procs:
        p.start()
    mic2, pics, mic3 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert mic2 == want
    assert len(pics) == 5
    for f in range(5):
        rc, pw = mico.pics_compress(stack[f], 4095, 3, 2)
        assert rc == 0 and pics[f] == pw
    rc, want3 = mico.wsi_compress(synth_slide())
    assert rc == 0 and mic3 == want3                                     # band by band + top of the pyramid = the slide coded in one piece

"""Static sharding of independent units over the GPUs of one node.

Strips (parallelstrips.go:77-93), MIC2 frames (multiframecompress.go:201-203) and MIC3 tiles (wsicompress.go:83-145) share no data,
so a rank codes a contiguous range of the units and the only cross-rank traffic is the assembly of a container on one rank:
an all-gather of the per-unit compressed sizes (8 bytes per unit) and a gather of the packed blobs -- point-to-point sends into
the slices of ONE buffer on the destination rank, exact sizes, no padding.  On the `nccl` backend (= RCCL) every tensor here is a
device tensor, so the blobs -- MIC3 tile payloads included -- go GPU to GPU over xGMI and never touch the host; on `gloo` (the CPU tests) the same code moves CPU
tensors.  Decode is the mirror image: the owner of a container scatters each rank's slice, every rank decodes its units.
No all-reduce, no collective inside the codec.  The codec is injected (a callable), so the CPU tests drive the plumbing with
gloo and the oracle, and the GPU tests / bench.py with a mic_hip session.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def shard_range(n_units: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous static partition: rank r owns units [n*r/world, n*(r+1)/world) (each rank's output is one
    contiguous blob range of the container)."""
    lo = (n_units * rank) // world
    hi = (n_units * (rank + 1)) // world
    return lo, hi


def _dist(group):
    import torch.distributed as dist
    return dist, dist.get_world_size(group), dist.get_rank(group)


def gather_unit_blobs(local_blobs, local_sizes, n_units_total: int, dst: int = 0, group=None):
    """local_blobs: uint8 tensor, this rank's units back to back; local_sizes: int64 tensor, one size per local unit (both on
    the backend's device).  Returns (all blobs in global unit order as ONE uint8 tensor on `dst`, None elsewhere;
    offsets[n_units_total + 1] as a numpy int64 array on every rank)."""
    import torch
    dist, world, rank = _dist(group)
    lo, hi = shard_range(n_units_total, world, rank)
    assert local_sizes.numel() == hi - lo, (local_sizes.numel(), lo, hi)
    dev = local_blobs.device
    per = max(shard_range(n_units_total, world, r)[1] - shard_range(n_units_total, world, r)[0] for r in range(world))
    mine = torch.zeros(max(per, 1), dtype=torch.int64, device=dev)
    mine[: hi - lo] = local_sizes.to(torch.int64)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)                                      # 1. sizes: 8 bytes per unit
    sizes = np.concatenate([parts[r][: shard_range(n_units_total, world, r)[1] - shard_range(n_units_total, world, r)[0]].cpu().numpy()
                            for r in range(world)]).astype(np.int64)
    offs = np.zeros(n_units_total + 1, dtype=np.int64)
    np.cumsum(sizes, out=offs[1:])
    shard_bytes = [int(offs[shard_range(n_units_total, world, r)[1]] - offs[shard_range(n_units_total, world, r)[0]]) for r in range(world)]
    assert local_blobs.numel() >= shard_bytes[rank]
    out = None
    if rank == dst:                                                                # 2. payloads: straight into their place
        out = torch.empty(int(offs[-1]), dtype=torch.uint8, device=dev)
        reqs = []
        for r in range(world):
            b0 = int(offs[shard_range(n_units_total, world, r)[0]])
            if r == rank:
                out[b0: b0 + shard_bytes[r]] = local_blobs[: shard_bytes[r]]
            elif shard_bytes[r]:
                reqs.append(dist.irecv(out[b0: b0 + shard_bytes[r]], src=_global_rank(dist, group, r), group=group))
        for q in reqs:
            q.wait()
    elif shard_bytes[rank]:
        dist.send(local_blobs[: shard_bytes[rank]].contiguous(), dst=_global_rank(dist, group, dst), group=group)
    return out, offs


def scatter_unit_blobs(all_blobs, offs: np.ndarray, n_units_total: int, src: int = 0, group=None, device=None):
    """The mirror image: `src` holds all blobs (global unit order) and the offsets; every rank receives the slice of its
    units.  Returns (local blobs tensor, local offsets rebased to 0)."""
    import torch
    dist, world, rank = _dist(group)
    meta = torch.zeros(n_units_total + 1, dtype=torch.int64, device=device if device is not None else (all_blobs.device if all_blobs is not None else "cpu"))
    if rank == src:
        meta.copy_(torch.from_numpy(np.asarray(offs, dtype=np.int64)))
    dist.broadcast(meta, src=_global_rank(dist, group, src), group=group)          # the offset table: 8 bytes per unit
    o = meta.cpu().numpy()
    lo, hi = shard_range(n_units_total, world, rank)
    nbytes = int(o[hi] - o[lo])
    local = torch.empty(nbytes, dtype=torch.uint8, device=meta.device)
    if rank == src:
        reqs = []
        for r in range(world):
            rlo, rhi = shard_range(n_units_total, world, r)
            piece = all_blobs[int(o[rlo]): int(o[rhi])]
            if r == rank:
                local.copy_(piece)
            elif piece.numel():
                reqs.append(dist.isend(piece.contiguous(), dst=_global_rank(dist, group, r), group=group))
        for q in reqs:
            q.wait()
    elif nbytes:
        dist.recv(local, src=_global_rank(dist, group, src), group=group)
    return local, (o[lo: hi + 1] - o[lo]).astype(np.int64)


def _global_rank(dist, group, r: int) -> int:
    return dist.get_global_rank(group, r) if group is not None else r


# ---- containers around gathered units ------------------------------------------------------------------------------------------
def mic2_header(width: int, height: int, sizes: Sequence[int]) -> bytes:
    """WriteMIC2 (multiframe.go:49-91), independent mode: 20-byte header + (offset, length) u32 per frame."""
    n = len(sizes)
    hdr = bytearray(20 + 8 * n)
    hdr[0:4] = b"MIC2"
    hdr[4:8] = int(width).to_bytes(4, "little"); hdr[8:12] = int(height).to_bytes(4, "little")
    hdr[12:16] = n.to_bytes(4, "little"); hdr[16] = 0x01
    off = 0
    for i, ln in enumerate(sizes):
        hdr[20 + 8 * i: 24 + 8 * i] = off.to_bytes(4, "little")
        hdr[24 + 8 * i: 28 + 8 * i] = int(ln).to_bytes(4, "little")
        off += int(ln)
    return bytes(hdr)


def write_mic2(width: int, height: int, blobs: Sequence[bytes]) -> bytes:
    return mic2_header(width, height, [len(b) for b in blobs]) + b"".join(blobs)


def pics_header(width: int, height: int, strip_h: int, sizes: Sequence[int]) -> bytes:
    """PICS (parallelstrips.go:20-28, :101-123): magic, width, height, strips, strip height, (offset, length) u32 per strip."""
    n = len(sizes)
    hdr = bytearray(20 + 8 * n)
    hdr[0:4] = b"PICS"
    for k, v in enumerate((width, height, n, strip_h)):
        hdr[4 + 4 * k: 8 + 4 * k] = int(v).to_bytes(4, "little")
    off = 0
    for i, ln in enumerate(sizes):
        hdr[20 + 8 * i: 24 + 8 * i] = off.to_bytes(4, "little")
        hdr[24 + 8 * i: 28 + 8 * i] = int(ln).to_bytes(4, "little")
        off += int(ln)
    return bytes(hdr)


# EncodeUnits: (first unit, one past the last unit) of the GLOBAL unit list -> (uint8 tensor of the blobs back to back,
# int64 tensor of their sizes), both on the backend's device.  DecodeUnits: (lo, hi, blobs tensor, offsets) -> pixels of the units.
EncodeUnits = Callable[[int, int], Tuple["object", "object"]]


def dist_compress_multi_frame(encode_units: EncodeUnits, width: int, height: int, n_frames_total: int, group=None) -> Optional[bytes]:
    """CompressMultiFrame (independent mode, multiframecompress.go:179-261) over the ranks: frame i is coded by the rank that
    owns it, rank 0 returns the MIC2 file (the others None)."""
    dist, world, rank = _dist(group)
    lo, hi = shard_range(n_frames_total, world, rank)
    blobs, sizes = encode_units(lo, hi)
    allb, offs = gather_unit_blobs(blobs, sizes, n_frames_total, dst=0, group=group)
    if rank != 0:
        return None
    return mic2_header(width, height, np.diff(offs)) + allb.cpu().numpy().tobytes()


def dist_decompress_multi_frame(decode_units, mic2: Optional[bytes], group=None, device=None):
    """DecompressMultiFrame over the ranks: rank 0 holds the file, every rank receives the streams of its frames and decodes
    them; returns (lo, hi, what decode_units returned for frames [lo, hi))."""
    import torch
    dist, world, rank = _dist(group)
    dev = device if device is not None else "cpu"
    head = torch.zeros(3, dtype=torch.int64, device=dev)
    allb, offs = None, None
    if rank == 0:
        w = int.from_bytes(mic2[4:8], "little"); h = int.from_bytes(mic2[8:12], "little"); n = int.from_bytes(mic2[12:16], "little")
        head.copy_(torch.tensor([w, h, n], dtype=torch.int64))
        tab = np.frombuffer(mic2, dtype="<u4", count=2 * n, offset=20).reshape(n, 2)
        offs = np.concatenate([tab[:, 0].astype(np.int64), [int(tab[-1, 0]) + int(tab[-1, 1])]])
        allb = torch.from_numpy(np.frombuffer(mic2, dtype=np.uint8, offset=20 + 8 * n).copy()).to(dev)
    dist.broadcast(head, src=_global_rank(dist, group, 0), group=group)
    w, h, n = (int(v) for v in head.cpu())
    local, loffs = scatter_unit_blobs(allb, offs, n, src=0, group=group, device=dev)
    lo, hi = shard_range(n, world, rank)
    return lo, hi, decode_units(lo, hi, local, loffs, w, h)


def dist_compress_pics_batch(encode_units: EncodeUnits, width: int, height: int, num_strips: int, n_frames_total: int, group=None) -> Optional[List[bytes]]:
    """A batch of frames, each a PICS file of num_strips strips (CompressParallelStrips, parallelstrips.go:55-124): frames are
    sharded over the ranks (units = strips, frame-major), rank 0 returns the PICS files."""
    dist, world, rank = _dist(group)
    ns = max(1, min(num_strips, height)); sh = (height + ns - 1) // ns; actual = (height + sh - 1) // sh
    flo, fhi = shard_range(n_frames_total, world, rank)
    blobs, sizes = encode_units(flo * actual, fhi * actual)
    # the unit partition must be the frame partition: gather per frame-shard
    allb, offs = _gather_by_shards(blobs, sizes, [(shard_range(n_frames_total, world, r)[0] * actual, shard_range(n_frames_total, world, r)[1] * actual)
                                                  for r in range(world)], group)
    if rank != 0:
        return None
    host = allb.cpu().numpy()
    files = []
    for f in range(n_frames_total):
        o = offs[f * actual: (f + 1) * actual + 1]
        files.append(pics_header(width, height, sh, np.diff(o)) + host[int(o[0]): int(o[-1])].tobytes())
    return files


def _gather_by_shards(blobs, sizes, ranges: Sequence[Tuple[int, int]], group):
    """gather_unit_blobs for an explicit (contiguous, ordered) unit range per rank"""
    import torch
    dist, world, rank = _dist(group)
    total = ranges[-1][1]
    dev = blobs.device
    per = max(1, max(hi - lo for lo, hi in ranges))
    mine = torch.zeros(per, dtype=torch.int64, device=dev)
    mine[: sizes.numel()] = sizes.to(torch.int64)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    allsz = np.concatenate([parts[r][: ranges[r][1] - ranges[r][0]].cpu().numpy() for r in range(world)]).astype(np.int64)
    offs = np.zeros(total + 1, dtype=np.int64)
    np.cumsum(allsz, out=offs[1:])
    nb = [int(offs[hi] - offs[lo]) for lo, hi in ranges]
    out = None
    if rank == 0:
        out = torch.empty(int(offs[-1]), dtype=torch.uint8, device=dev)
        reqs = []
        for r, (lo, hi) in enumerate(ranges):
            if r == 0:
                out[: nb[0]] = blobs[: nb[0]]
            elif nb[r]:
                reqs.append(dist.irecv(out[int(offs[lo]): int(offs[lo]) + nb[r]], src=_global_rank(dist, group, r), group=group))
        for q in reqs:
            q.wait()
    elif nb[rank]:
        dist.send(blobs[: nb[rank]].contiguous(), dst=_global_rank(dist, group, 0), group=group)
    return out, offs


# ---- the injected codec on a GPU: a mic_hip session over a device-resident unit array ---------------------------------------
def session_codec(mic, sess, d_pixels, units: Sequence[Tuple[int, int, int, int, int]]):
    """EncodeUnits / DecodeUnits over `units` = the GLOBAL list (px_offset relative to this rank's d_pixels for the units it
    owns).  Blobs come back as a device tensor copied out of the session (the session reuses its packed buffer)."""
    import torch

    def encode(lo: int, hi: int):
        cu = mic.Session.make_units(list(units[lo:hi]))
        sess.encode_enqueue(d_pixels.data_ptr(), cu)
        d_blobs, offs, st, _ = sess.encode_finish()
        if (st != 0).any():
            raise mic.MicError(int(st[st != 0][0]), "session_codec.encode")
        out = torch.empty(int(offs[-1]), dtype=torch.uint8, device=d_pixels.device)
        mic.device_copy(out.data_ptr(), d_blobs, int(offs[-1]))
        return out, torch.from_numpy(np.diff(offs.astype(np.int64))).to(d_pixels.device)

    def decode(lo: int, hi: int, blobs, offs, w: int, h: int):
        cu = mic.Session.make_units([(i * w * h, w, h, 0, 0) for i in range(hi - lo)])
        out = torch.empty((hi - lo, h, w), dtype=torch.int16, device=blobs.device)
        sess.decode_enqueue(blobs.data_ptr(), offs.astype(np.uint64), cu, out.data_ptr())
        st = sess.decode_finish()
        if (st != 0).any():
            raise mic.MicError(int(st[st != 0][0]), "session_codec.decode")
        return out

    return encode, decode


# ---- MIC3: a slide in bands of tile rows -------------------------------------------------------------------------------------
# CompressWSI (wsicompress.go:83-145) codes every tile of every pyramid level on its own.  Level-0 tiles are three quarters of the
# work and split into bands of tile rows with no shared data; a level-k tile covers 2^k tile rows of level 0, so a band that is a
# multiple of tile_h * 2^K rows holds whole tiles of levels 0..K and the box filter (Downsample2xRGB, wsipyramid.go:10-32: 2x2
# blocks, floor) never reaches across its edge.  Each rank therefore codes levels 0..K of its band as a slide of its own; the
# rows of level K+1 (1 / 4^(K+1) of the pixels) are gathered on rank 0, which codes the few top levels, and the tile blobs are
# gathered like any other unit.  The container is the reference's (wsiformat.go:99-285): 48-byte header, 20 bytes per level,
# 16 bytes per tile, blobs in level order.
def wsi_levels(width: int, height: int, tile_w: int, tile_h: int, levels_req: int = 0):
    """autoLevelCount + computeLevels (wsiformat.go:244-285) with the pyramid's own stop (a level of zero rows or columns is
    not built): [(level width, level height, tiles across, tiles down)]"""
    n = levels_req
    if n <= 0:
        n, ww, hh = 1, width, height
        while ww > tile_w or hh > tile_h:
            ww //= 2; hh //= 2; n += 1
            if ww <= 1 and hh <= 1:
                break
    out, ww, hh = [], width, height
    for i in range(n):
        if i and (ww == 0 or hh == 0):
            break
        out.append((ww, hh, (ww + tile_w - 1) // tile_w, (hh + tile_h - 1) // tile_h))
        ww //= 2; hh //= 2
    return out


def wsi_band_plan(height: int, tile_h: int, num_levels: int, world: int):
    """(K, [(y0, y1) per rank]): bands of tile_h * 2^K rows, K the largest level that still leaves every rank a band"""
    k = num_levels - 1
    while k > 0 and (height + (tile_h << k) - 1) // (tile_h << k) < world:
        k -= 1
    a = tile_h << k
    nblocks = (height + a - 1) // a
    bands = []
    for r in range(world):
        lo, hi = shard_range(nblocks, world, r)
        bands.append((min(height, lo * a), min(height, hi * a)))
    return k, bands


def parse_mic3(buf: bytes):
    """-> ([(lw, lh, ltx, lty, first tile)], tile lengths (int64 array, container order), offset of the first blob)"""
    assert buf[:4] == b"MIC3"
    nlev = int.from_bytes(buf[28:30], "little"); total = int.from_bytes(buf[32:40], "little")
    lv = [tuple(int.from_bytes(buf[48 + 20 * i + 4 * k: 52 + 20 * i + 4 * k], "little") for k in range(5)) for i in range(nlev)]
    tab = np.frombuffer(buf, dtype="<u8", count=2 * total, offset=48 + 20 * nlev).reshape(total, 2)
    return lv, tab[:, 1].astype(np.int64), 48 + 20 * nlev + 16 * total


def mic3_header(width: int, height: int, tile_w: int, tile_h: int, channels: int, bits: int, levels, sizes: Sequence[int]) -> bytes:
    """WriteMIC3 header, level table and tile index (wsiformat.go:99-190) for tiles of the given sizes in container order"""
    total = len(sizes)
    hdr = bytearray(48 + 20 * len(levels) + 16 * total)
    hdr[0:4] = b"MIC3"; hdr[4:8] = (1).to_bytes(4, "little")
    for k, v in enumerate((width, height, tile_w, tile_h)):
        hdr[8 + 4 * k: 12 + 4 * k] = int(v).to_bytes(4, "little")
    hdr[24] = channels; hdr[26] = bits; hdr[27] = 0x01 | (0x02 if channels == 3 else 0)
    hdr[28:30] = len(levels).to_bytes(2, "little"); hdr[32:40] = total.to_bytes(8, "little")
    first = 0
    for i, (lw, lh, ltx, lty) in enumerate(levels):
        for k, v in enumerate((lw, lh, ltx, lty, first)):
            hdr[48 + 20 * i + 4 * k: 52 + 20 * i + 4 * k] = int(v).to_bytes(4, "little")
        first += ltx * lty
    off, base = 0, 48 + 20 * len(levels)
    for i, ln in enumerate(sizes):
        hdr[base + 16 * i: base + 16 * i + 8] = off.to_bytes(8, "little")
        hdr[base + 16 * i + 8: base + 16 * i + 16] = int(ln).to_bytes(8, "little")
        off += int(ln)
    return bytes(hdr)


def downsample2x(img):
    """Downsample2xRGB / Downsample2xGrey (wsipyramid.go:10-55) on a (rows, cols[, channels]) integer torch tensor"""
    import torch
    nh, nw = img.shape[0] // 2, img.shape[1] // 2
    v = img[: 2 * nh, : 2 * nw].to(torch.int32)
    return ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) // 4).to(img.dtype)


def dist_compress_wsi(encode_slide, band, width: int, height: int, tile_w: int = 256, tile_h: int = 256, levels_req: int = 0,
                      channels: int = 3, bits: int = 8, group=None, out: Optional[np.ndarray] = None):
    """CompressWSI over the ranks.  band: this rank's rows of the slide (wsi_band_plan's (y0, y1)) as a (rows, width[, channels])
    tensor on the backend's device; encode_slide(image tensor, levels) -> (payload: uint8 tensor on the backend's device holding
    every tile blob of that image in container order, tile sizes: int64 numpy array, level table [(w, h, tiles across, tiles
    down)]) -- the injected codec: a mic_hip session on a GPU (session_wsi_codec: the payload never leaves the device), the
    oracle in the CPU tests (slide_codec_from_bytes).  Rank 0 returns the slide's MIC3 file as bytes, the others None; with
    `out` (a uint8 array of the caller's -- pinned memory makes the one device-to-host copy a plain DMA) rank 0 writes the file
    there and returns its length."""
    import torch
    dist, world, rank = _dist(group)
    levels = wsi_levels(width, height, tile_w, tile_h, levels_req)
    L = len(levels)
    K, bands = wsi_band_plan(height, tile_h, L, world)
    y0, y1 = bands[rank]
    assert band.shape[0] == y1 - y0 and band.shape[1] == width
    dev = band.device
    # 1. levels 0..K of the band, a slide of its own
    nloc = [[((b1 - b0) >> k) for k in range(K + 1)] for b0, b1 in bands]                    # band rows per level (floor chain = shift: aligned bands)
    tiles_of = [[((rows + tile_h - 1) // tile_h) * levels[k][2] if rows > 0 and k < L else 0 for k, rows in enumerate(nl)] for nl in nloc]
    payload, sizes = torch.empty(0, dtype=torch.uint8, device=dev), torch.empty(0, dtype=torch.int64, device=dev)
    if y1 > y0:
        payload, sz, lv = encode_slide(band, min(K + 1, L))
        assert [t[2] * t[3] for t in lv] == [t for t in tiles_of[rank] if t], (lv, tiles_of[rank])
        sizes = torch.from_numpy(np.asarray(sz, dtype=np.int64)).to(dev)
    starts = np.concatenate([[0], np.cumsum([sum(t) for t in tiles_of])]).astype(np.int64)
    allb, offs = _gather_by_shards(payload, sizes, [(int(starts[r]), int(starts[r + 1])) for r in range(world)], group)
    # 2. the rows of level K + 1 go to rank 0, which codes the top of the pyramid
    top = None
    if L > K + 1:
        t = band
        for _ in range(K + 1):
            t = downsample2x(t)
        flat = t.contiguous().view(torch.uint8).reshape(-1)
        row_bytes = levels[K + 1][0] * channels * (2 if bits == 16 else 1)
        nrows = [(b1 - b0) >> (K + 1) for b0, b1 in bands]
        topb, _ = _gather_by_shards(flat, torch.tensor([flat.numel()], dtype=torch.int64, device=dev), [(r, r + 1) for r in range(world)], group)
        if rank == 0:
            assert topb.numel() == sum(nrows) * row_bytes == levels[K + 1][1] * row_bytes, (topb.numel(), nrows, levels[K + 1])
            shape = (levels[K + 1][1], levels[K + 1][0]) + ((channels,) if channels > 1 else ())
            top = encode_slide(topb.view(band.dtype).reshape(shape), L - K - 1)
    if rank != 0:
        return None
    # 3. the container: level by level, band by band (slices of the gathered device buffer, one concatenation, one copy to the host)
    pieces, out_sizes = [], []
    for k in range(min(K + 1, L)):
        for r in range(world):
            t0 = int(starts[r]) + sum(tiles_of[r][:k]); t1 = t0 + tiles_of[r][k]
            pieces.append(allb[int(offs[t0]): int(offs[t1])]); out_sizes.extend(np.diff(offs[t0: t1 + 1]).tolist())
    if top is not None:
        tp, tsz, tlv = top
        assert [tuple(a) for a in tlv] == [tuple(a) for a in levels[K + 1:]], (tlv, levels[K + 1:])
        pieces.append(tp); out_sizes.extend(np.asarray(tsz).tolist())
    assert len(out_sizes) == sum(a[2] * a[3] for a in levels)
    body = torch.cat(pieces) if len(pieces) != 1 else pieces[0]
    hdr = mic3_header(width, height, tile_w, tile_h, channels, bits, levels, out_sizes)
    if out is None:
        return hdr + body.cpu().numpy().tobytes()
    total = len(hdr) + body.numel()
    assert out.dtype == np.uint8 and out.size >= total, "dist_compress_wsi: out is too small"
    out[: len(hdr)] = np.frombuffer(hdr, dtype=np.uint8)
    torch.from_numpy(out[len(hdr): total]).copy_(body)                       # one device-to-host copy, straight into the caller's buffer
    return total


def slide_codec_from_bytes(encode_file, device="cpu"):
    """encode_slide for dist_compress_wsi from a function that returns a MIC3 FILE as bytes (the oracle in the CPU tests)"""
    import torch

    def encode_slide(img, levels: int):
        f = encode_file(img, levels)
        lv, sz, d0 = parse_mic3(f)
        payload = torch.from_numpy(np.frombuffer(f, dtype=np.uint8, offset=d0).copy()).to(device)
        return payload, sz, [(a, b, c, d) for a, b, c, d, _ in lv]
    return encode_slide


def session_wsi_codec(mic, sess, tile_w: int = 256, tile_h: int = 256):
    """encode_slide for dist_compress_wsi on a GPU: the band (a device tensor) goes through the session's device-resident MIC3
    encoder, a kernel lays the coded planes out as the container's payload (mic_hip_session_wsi_payload), and that payload is
    handed on as a device tensor: nothing but the tile sizes touches the host"""
    import torch

    def encode_slide(img, levels: int):
        img = img.contiguous()
        ch = img.shape[2] if img.dim() == 3 else 1
        bits = 16 if img.element_size() == 2 else 8
        w, h = int(img.shape[1]), int(img.shape[0])
        tiles, _ = sess.wsi_encode(img.data_ptr(), w, h, ch, bits, tile_w, tile_h, levels)
        d_payload, nbytes, lens = sess.wsi_payload(tiles)
        out = torch.empty(nbytes, dtype=torch.uint8, device=img.device)
        if nbytes:
            mic.device_copy(out.data_ptr(), d_payload, nbytes)
        lv = [(lw, lh, (lw + tile_w - 1) // tile_w, (lh + tile_h - 1) // tile_h) for lw, lh in sess.wsi_levels()]
        return out, lens, lv
    return encode_slide

"""Rehearsal of bench.py's one-process multi-device end_to_end leg on a one-GPU box: the same device listed twice (two shards, two
sessions, one PCIe link), sixteen frames.  usage: python tools/rehearse_e2e.py"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, __graft_entry__ as entry
import torch
mic = entry.load_package()
synth = importlib.import_module("medical_image_codec_amd.synth")
dev = torch.device("cuda:0")
d_px = synth.xr_like_batch_torch(16, cols=2577, rows=2048, depth=12, seed0=1, noise=synth.XR_NOISE_PUBLISHED_RATIO, device=dev)
r = bench.leg_end_to_end(mic, torch, d_px, 2577, 2048, 8, 4095, dev, devices=[0, 0])
print(json.dumps(r)[:900])

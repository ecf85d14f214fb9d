"""Runs single legs of bench.py (configs 3 / 4 / 5) and prints their result: python tools/bench_legs.py wavelet|mic2|wsi [args]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
mic = bench.entry.load_package()
synth = importlib.import_module("medical_image_codec_amd.synth")
mic.lib().mic_hip_set_device(0)
dev = torch.device("cuda:0")
which = sys.argv[1]
if which == "wavelet":
    r = bench.leg_wavelet(mic, torch, synth, dev, 2, 1, nframes=int(sys.argv[2]) if len(sys.argv) > 2 else 256)
elif which == "mic2":
    r = bench.leg_mic2(mic, torch, synth, dev, 3, 1)
else:
    r = bench.leg_wsi(mic, torch, synth, dev, 2, size=int(sys.argv[2]) if len(sys.argv) > 2 else 32768)
print(json.dumps(r, indent=1))

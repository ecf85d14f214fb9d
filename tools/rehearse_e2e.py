import sys, os, importlib, json
sys.path.insert(0, "/root/repo")
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.getcwd())
import bench, __graft_entry__ as entry, torch
mic = entry.load_package()
synth = importlib.import_module("medical_image_codec_amd.synth")
dev = torch.device("cuda:0")
d_px = synth.xr_like_batch_torch(16, cols=2577, rows=2048, depth=12, seed0=1, noise=synth.XR_NOISE_PUBLISHED_RATIO, device=dev)
r = bench.leg_end_to_end(mic, torch, d_px, 2577, 2048, 8, 4095, dev, devices=[0, 0])
print(json.dumps(r)[:900])

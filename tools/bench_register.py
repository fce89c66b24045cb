"""Latency of one registration iteration (configs[3]: 512^3 CT, one pose, 256^2 then 512^2 detector):
render -> XrayTransforms -> mNCC + gradNCC -> backward -> Adam.  Run on the GPU box.  (The measuring code is tools/benchlib.py,
which bench.py's `c4_register_ms_per_pose_iteration` leg runs as well; this script adds the configurations beyond the default.)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import benchlib  # noqa: E402
from xvr_amd.data import make_phantom, read  # noqa: E402

dev = torch.device("cuda")
vol, _ = make_phantom(512, n_ellipsoids=64, seed=0, device=dev)
extra = (dict(sigma=1.0), dict(equalize=True), dict(sigma=1.0, equalize=True), dict(parameterization="se3_log_map"),
         dict(parameterization="axis_angle"), dict(parameterization="quaternion"), dict(parameterization="quaternion_adjugate"),
         dict(parameterization="rotation_6d"), dict(use_graph=False))
res = benchlib.c4_register(dev, read(vol, orientation="AP"), extra=extra)
for H, e in res.items():
    print(f"detector {H}^2: {e['single']:.3f} ms / iteration (one start, ncc {e['ncc'][0]:.4f} -> {e['ncc'][1]:.4f}); "
          f"8 starts as one batch: {e['batched8']:.3f} ms per pose-iteration")
    for k, v in e.items():
        if k.startswith("single "):
            print(f"  Registrar.run {k[7:]:46s} {v:.3f} ms / iteration")
    for k, v in sorted(e["kernels"].items(), key=lambda kv: -kv[1]["avg_ms"]):
        print(f"    {k:30s} {v['launches']:4d} x {v['avg_ms'] * 1e3:7.1f} us")

"""The render-side of one xvr training step (configs[4]; /root/reference/src/xvr/model/trainer.py:185-230) at full
size: HU -> density with a random bone multiplier, render #1 (no grad, 8 label channels), render #2 at the predicted
poses with the pose gradient, PoseRegressionLoss and its backward.  The timm regressor is out of scope: the
"network output" is a leaf tensor of pose parameters.  Run on the GPU box.  (The measuring code is tools/benchlib.py, which
bench.py's `c5_train_step` leg runs as well.)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import benchlib  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
r = benchlib.c5_train_step(torch.device("cuda"), size=size)
print(f"training step (render side), {r['config']}: {r['ms_per_step']:.2f} ms")
for k, v in r["phases"].items():
    print(f"  {k:38s} {v:7.2f} ms")
for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"])[:14]:
    print(f"    {k:36s} {v['launches']:4d} x {v['avg_ms']:7.3f} ms")

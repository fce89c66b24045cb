"""Timeline of the 16^3-brick splat at the benchmark's size: builds a diagnostic copy of the library (-DXVR_S16_TRACE)
whose kernel records, per workgroup, wall clock at start / end, visits and samples; prints how full the machine is over
the launch and how the work is spread over the bricks.  Run on the GPU box:  python tools/splat_trace.py"""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from xvr_amd.build import build_diagnostic_library  # noqa: E402

extra = sys.argv[1:]   # further -D's, e.g. XVR_S16_TAB=512 XVR_S16_WAVES=5
os.environ["XVR_DRR_LIBRARY"] = str(build_diagnostic_library(["XVR_S16_TRACE"] + extra, ROOT / "tools" / "_build" / "libxvr_drr_s16trace.so"))

import torch  # noqa: E402

from xvr_amd import _lib  # noqa: E402
from xvr_amd.data import make_phantom, read  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402
from xvr_amd.training import get_random_pose  # noqa: E402

dev = torch.device("cuda")
vol, _ = make_phantom(512, n_ellipsoids=64, seed=0, device=dev)
drr = DRR(read(vol, orientation="AP"), 1020.0, 256, 1.08821875, renderer="trilinear", reverse_x_axis=False).to(dev)
pose = get_random_pose(135.0, 225.0, -45.0, 45.0, -15.0, 15.0, -150.0, 150.0, 450.0, 1000.0, -150.0, 150.0, 116,
                       generator=torch.Generator().manual_seed(0)).to(dev)
density = drr.density.clone().requires_grad_()
raw = ctypes.CDLL(str(_lib.library_path()))
for _ in range(3):
    density.grad = None
    drr(pose, density=density).sum().backward()
torch.cuda.synchronize()
n = 32768
out = (ctypes.c_ulonglong * (12 * n))()
assert raw.xvr_drr_debug_s16_trace(out, n) == 0
print("workgroups per CU according to hipOccupancyMaxActiveBlocksPerMultiprocessor:", raw.xvr_drr_debug_s16_occupancy())
t = np.frombuffer(out, dtype=np.uint64).reshape(n, 12).astype(np.float64)
t0, t1, visits, samples = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
tick = 1e-8   # wall_clock64: 100 MHz
t0 = (t0 - t0.min()) * tick * 1e3
t1 = (t1 - t[:, 0].min()) * tick * 1e3
dur = t1 - t0
print(f"launch {t1.max():.2f} ms; workgroups {n}, with work {(visits > 0).sum()}; visits {visits.sum():.4g}, runs {samples.sum():.4g}")
print(f"workgroup-time {dur.sum():.1f} ms = {dur.sum() / t1.max():.0f} workgroups resident on average (1024 = 4 per CU)")
w = visits > 0
print(f"per visit: {1e3 * dur[w].sum() / visits.sum():.2f} us, {samples.sum() / visits.sum():.0f} runs")
print("duration of a workgroup with work: median %.3f ms, 90 %% %.3f, 99 %% %.3f, max %.3f" % tuple(np.percentile(dur[w], [50, 90, 99, 100])))
tk = t[:, 4:10].sum(axis=0)
names = ["pose set-up", "enumeration", "wait: list barrier", "splat loop", "wait: cells barrier", "flush"]
print("wavefront 0's shader clocks per visit: " + ", ".join(f"{nm} {v / visits.sum():.0f}" for nm, v in zip(names, tk)) + f" (sum {tk.sum() / visits.sum():.0f})")
print(f"visits with an empty list: {t[:, 10].sum() / visits.sum():.1%}; with fewer than 32 runs: {t[:, 11].sum() / visits.sum():.1%}")
edges = np.linspace(0, t1.max(), 8)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    res = ((t0 <= mid) & (t1 > mid)).sum()
    print(f"  t = {mid:6.2f} ms: {res:5d} workgroups resident, {((t0 >= a) & (t0 < b)).sum():6d} started in the bin")

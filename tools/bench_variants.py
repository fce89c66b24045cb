"""Forward + backward (pose AND voxel gradient) at the benchmark's size for every RenderSpec variant the parity tests
cover (tests/test_hip_parity.py SPECS) plus the masked renders -- the knobs parity against the real diffdrr may land on
(DESIGN.md section 2).  One markdown table: ms per step, DRRs/s, and the kernels of the step with their HIP-event times.
Run on the GPU box:  python tools/bench_variants.py [--batch 116 --size 512 --det 256] > profiles/r03_variants.md"""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import deepfluoro_poses  # noqa: E402
from xvr_amd import renderers  # noqa: E402
from xvr_amd.data import make_phantom, read  # noqa: E402
from xvr_amd.drr import DRR  # noqa: E402

VARIANTS = [
    ("trilinear (default)", "trilinear", dict(), dict(n_points=500), None),
    ("trilinear voxel_shift=0, /(N-1)", "trilinear", dict(voxel_shift=0.0, step_mode="n_minus_1"), dict(n_points=500), None),
    ("trilinear norm_dims_offset=-1", "trilinear", dict(norm_dims_offset=-1), dict(n_points=500), None),
    ("trilinear align_corners, offset -1, shift 0", "trilinear", dict(voxel_shift=0.0, norm_dims_offset=-1), dict(n_points=500, align_corners=True), None),
    ("trilinear near=0.2 far=0.9", "trilinear", dict(near=0.2, far=0.9), dict(n_points=500), None),
    ("trilinear clip_to_volume", "trilinear", dict(clip_to_volume=True), dict(n_points=500), None),
    ("trilinear clip_to_volume='batch' (one alpha window per call)", "trilinear", dict(clip_to_volume="batch"), dict(n_points=500), None),
    ("trilinear mask -> 8 channels, summed (xvr)", "trilinear", dict(), dict(n_points=500), "sum"),
    ("trilinear mask -> 8 channels, per-channel gradient", "trilinear", dict(), dict(n_points=500), "per-channel"),
    ("siddon (default)", "siddon", dict(), dict(), None),
    ("siddon voxel_shift=0", "siddon", dict(voxel_shift=0.0), dict(), None),
    ("siddon norm_dims_offset=+1 (not the exact index map)", "siddon", dict(norm_dims_offset=1), dict(), None),
    ("siddon mask -> 8 channels, summed (xvr)", "siddon", dict(), dict(), "sum"),
    ("siddon mask -> 8 channels, per-channel gradient", "siddon", dict(), dict(), "per-channel"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=116)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--det", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--slow-steps", type=int, default=1, help="timed steps for variants slower than 100 ms")
    ap.add_argument("--only", default="", help="run only the variants whose name contains this text")
    args = ap.parse_args()
    dev = torch.device("cuda")
    B, H = args.batch, args.det
    vol, lab = make_phantom(args.size, n_ellipsoids=64, n_labels=8, seed=0, device=dev)
    rot, xyz = (t.to(dev) for t in deepfluoro_poses(B, seed=0).convert("euler_angles", "ZXY"))
    print(f"# fwd + bwd(pose + voxel) for every RenderSpec variant, {args.size}^3 -> {H}^2, batch {B} (tools/bench_variants.py)\n")
    print("| variant | ms / step | DRRs/s | kernels (HIP-event ms) |\n|---|---|---|---|")
    for name, renderer, ctor_kw, call_kw, masked in VARIANTS:
        if args.only and args.only not in name:
            continue
        ctor_kw = dict(ctor_kw)
        voxel_shift = ctor_kw.pop("voxel_shift", 0.5)
        sub = read(vol, lab if masked else None, orientation="AP")
        drr = DRR(sub, 1020.0, H, 1.08821875 * 256 / H, renderer=renderer, reverse_x_axis=False, voxel_shift=voxel_shift, **ctor_kw).to(dev)
        density = drr.density.clone().requires_grad_(True)
        r, t = rot.clone().requires_grad_(True), xyz.clone().requires_grad_(True)
        C = 8 if masked else 1
        w = torch.rand(B, C if masked == "per-channel" else 1, H, H, device=dev)

        def step():
            density.grad = r.grad = t.grad = None
            img = drr(r, t, parameterization="euler_angles", convention="ZXY", density=density, mask_to_channels=bool(masked), **call_kw)
            if masked == "sum":
                img = img.sum(dim=1, keepdim=True)
            (img * w).sum().backward()

        step()
        step()   # (the y-pair / bricked copy of the volume is built on the third render of a volume version: keep it out of the timing)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        n = args.steps if first < 0.1 else args.slow_steps
        renderers.PROFILER = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        ev, renderers.PROFILER = renderers.PROFILER, None
        per = {}
        for k, e0, e1 in ev:
            per.setdefault(k, []).append(e0.elapsed_time(e1))
        kern = ", ".join(f"{k} {sum(v) / len(v):.2f}" for k, v in per.items() if sum(v) / len(v) >= 0.05)
        print(f"| {name} | {dt * 1e3:.1f} | {B / dt:.0f} | {kern} |", flush=True)
        del drr, density


if __name__ == "__main__":
    main()

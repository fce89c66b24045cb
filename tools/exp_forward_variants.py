"""Trilinear forward (+ jacobian) at the benchmark's size under tuning builds of drr_trilinear.hip, with a bit-for-bit check
of every variant against the product library.    build | run | one <tag>

Variants: the software-pipelined march (-DXVR_FWD_PIPE_U=<samples per trip> [-DXVR_FWD_PIPE_ASM=1]) x occupancy cap
(-DXVR_FWD_WAVES).  Diagnostics with the product library: the natural layout, and 116 copies of ONE pose (the whole
launch's footprint fits the 256 MiB Infinity Cache: what the same kernel costs when no tap ever comes from HBM).
"""
import hashlib
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

VARIANTS = {   # tag -> defines
    "p1a_w4": ["XVR_FWD_PIPE_U=1", "XVR_FWD_PIPE_ASM=1"],
    "p2a_w4": ["XVR_FWD_PIPE_U=2", "XVR_FWD_PIPE_ASM=1"],
    "p2c_w4": ["XVR_FWD_PIPE_U=2"],
    "p2a_w3": ["XVR_FWD_PIPE_U=2", "XVR_FWD_PIPE_ASM=1", "XVR_FWD_WAVES=3"],
    "p3a_w3": ["XVR_FWD_PIPE_U=3", "XVR_FWD_PIPE_ASM=1", "XVR_FWD_WAVES=3"],
    "p4a_w3": ["XVR_FWD_PIPE_U=4", "XVR_FWD_PIPE_ASM=1", "XVR_FWD_WAVES=3"],
    "p4a_w2": ["XVR_FWD_PIPE_U=4", "XVR_FWD_PIPE_ASM=1", "XVR_FWD_WAVES=2"],
    "p1a_w6": ["XVR_FWD_PIPE_U=1", "XVR_FWD_PIPE_ASM=1", "XVR_FWD_WAVES=6"],
    "p1a_w8": ["XVR_FWD_PIPE_U=1", "XVR_FWD_PIPE_ASM=1", "XVR_FWD_WAVES=8"],
}


def lib(tag):
    from xvr_amd.build import diagnostic_path
    return diagnostic_path(f"fwd_{tag}")


def one(mode):
    import torch
    from bench import deepfluoro_poses
    from xvr_amd import renderers
    from xvr_amd.data import make_phantom, read
    from xvr_amd.drr import DRR

    dev = torch.device("cuda")
    B, H = 116, 256
    rot, xyz = (t.to(dev) for t in deepfluoro_poses(B, seed=0).convert("euler_angles", "ZXY"))
    if mode == "samepose":
        rot, xyz = rot[:1].expand(B, 3).contiguous(), xyz[:1].expand(B, 3).contiguous()
    if mode == "natural":
        renderers.YPAIR_LAYOUT = False
    vol, _ = make_phantom(512, n_ellipsoids=64, seed=0, device=dev)
    drr = DRR(read(vol, orientation="AP"), 1020.0, H, 1.08821875, renderer="trilinear", reverse_x_axis=False).to(dev)
    rot.requires_grad_(True)
    xyz.requires_grad_(True)
    w = torch.rand(B, 1, H, H, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    for _ in range(3):
        img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY", n_points=500)
    renderers.PROFILER = []
    for _ in range(6):
        rot.grad = xyz.grad = None
        img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY", n_points=500)
        (img * w).sum().backward()
    torch.cuda.synchronize()
    ev, renderers.PROFILER = renderers.PROFILER, None
    t = [a.elapsed_time(b) for k, a, b in ev if "forward" in k]
    h = hashlib.sha1()
    for x in (img, rot.grad, xyz.grad):
        h.update(x.detach().cpu().numpy().tobytes())
    print(json.dumps({"ms": sum(t) / len(t), "min_ms": min(t), "sha": h.hexdigest()[:16]}))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        from xvr_amd.build import build_diagnostic_library
        for tag, defs in VARIANTS.items():
            print(build_diagnostic_library(defs, lib(tag), only=["drr_trilinear.hip"]), flush=True)
    elif sys.argv[1] == "one":
        one(sys.argv[2])
    else:
        runs = [("product", None, "default"), ("product, natural layout", None, "natural"), ("product, 116 x one pose", None, "samepose")]
        runs += [(tag, lib(tag), "default") for tag in VARIANTS]
        ref = None
        for name, path, mode in runs:
            env = dict(os.environ)
            if path is not None:
                if not path.exists():
                    continue
                env["XVR_DRR_LIBRARY"] = str(path)
            out = subprocess.run([sys.executable, __file__, "one", mode], env=env, capture_output=True, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
            except (IndexError, ValueError):
                print(f"{name}: FAILED\n{out.stderr[-2000:]}", flush=True)
                continue
            if name == "product":
                ref = d["sha"]
            same = "" if mode != "default" else ("  bits identical" if d["sha"] == ref else "  BITS DIFFER")
            print(f"{name:28s} forward+jac {d['ms']:.3f} ms (min {d['min_ms']:.3f}){same}", flush=True)

"""Trilinear forward (+ jacobian) at the benchmark's size (512^3 -> 256^2, 116 poses, n_points 500) under the launch options of
the slab-major march (HISTORY.md section 4.4), with two diagnostics.  Run on the GPU box:   run | one <mode>

Diagnostics with the one-launch march: the natural layout, and 116 copies of ONE pose (the whole launch's footprint fits the
256 MiB Infinity Cache: what the same kernel costs when no tap ever comes from HBM).
"""
import hashlib
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def one(mode):
    import torch
    from bench import deepfluoro_poses
    from xvr_amd import renderers
    from xvr_amd.data import make_phantom, read
    from xvr_amd.drr import DRR

    dev = torch.device("cuda")
    B, H = 116, 256
    rot, xyz = (t.to(dev) for t in deepfluoro_poses(B, seed=0).convert("euler_angles", "ZXY"))
    if "samepose" in mode:
        rot, xyz = rot[:1].expand(B, 3).contiguous(), xyz[:1].expand(B, 3).contiguous()
    if "sortyaw" in mode:     # poses ordered by view direction: neighbours in the launch look along neighbouring directions
        order = torch.argsort(rot[:, 0])
        rot, xyz = rot[order].contiguous(), xyz[order].contiguous()
    if "sortdir" in mode:     # ... by yaw in 4 bands, by pitch within a band (a snake through the two angles)
        band = torch.clamp(((rot[:, 0] - rot[:, 0].min()) / (rot[:, 0].max() - rot[:, 0].min() + 1e-6) * 4).long(), max=3)
        key = band.double() * 10 + torch.where(band % 2 == 0, rot[:, 1], -rot[:, 1]).double()
        order = torch.argsort(key)
        rot, xyz = rot[order].contiguous(), xyz[order].contiguous()
    if "natural" in mode:
        renderers.YPAIR_LAYOUT = False
    vol, _ = make_phantom(512, n_ellipsoids=64, seed=0, device=dev)
    drr = DRR(read(vol, orientation="AP"), 1020.0, H, 1.08821875, renderer="trilinear", reverse_x_axis=False).to(dev)
    rot.requires_grad_("nojac" not in mode)
    w = torch.rand(B, 1, H, H, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    for _ in range(3):
        img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY", n_points=500)
    renderers.PROFILER = []
    for _ in range(6):
        rot.grad = None
        img = drr(rot, xyz, parameterization="euler_angles", convention="ZXY", n_points=500)
        if rot.requires_grad:
            (img * w).sum().backward()
    torch.cuda.synchronize()
    ev, renderers.PROFILER = renderers.PROFILER, None
    t = [a.elapsed_time(b) for k, a, b in ev if k.startswith("trilinear_forward")]
    h = hashlib.sha1()
    for x in (img, rot.grad) if rot.requires_grad else (img,):
        h.update(x.detach().cpu().numpy().tobytes())
    print(json.dumps({"ms": sum(t) / len(t), "min_ms": min(t), "sha": h.hexdigest()[:16], "sum": img.double().sum().item()}))


if __name__ == "__main__":
    if sys.argv[1] == "one":
        one(sys.argv[2])
    elif sys.argv[1] == "tiles":   # round 4: workgroup tile geometry (option tile_geom) and the order of the poses in the launch
        for name, mode, env in [("16x16 tiles (default)", "default", {"XVR_DRR_TILE_GEOM": "0"}),
                                ("8x32 tiles, long along z", "default", {"XVR_DRR_TILE_GEOM": "1"}),
                                ("4x64 tiles, long along z", "default", {"XVR_DRR_TILE_GEOM": "2"}),
                                ("16x16, poses sorted by yaw", "sortyaw", {"XVR_DRR_TILE_GEOM": "0"}),
                                ("16x16, poses sorted yaw band / pitch", "sortdir", {"XVR_DRR_TILE_GEOM": "0"}),
                                ("8x32, poses sorted by yaw", "sortyaw", {"XVR_DRR_TILE_GEOM": "1"}),
                                ("8x32, natural layout", "natural", {"XVR_DRR_TILE_GEOM": "1"}),
                                ("4x64, natural layout", "natural", {"XVR_DRR_TILE_GEOM": "2"}),
                                ("8x32, no jacobian", "nojac", {"XVR_DRR_TILE_GEOM": "1"}),
                                ("16x16, no jacobian", "nojac", {"XVR_DRR_TILE_GEOM": "0"})]:
            out = subprocess.run([sys.executable, __file__, "one", mode], env=dict(os.environ, XVR_DRR_FWD_SLABS="0", **env), capture_output=True, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                print(f"{name:40s} forward {d['ms']:.3f} ms (min {d['min_ms']:.3f})  sha {d['sha']}", flush=True)
            except (IndexError, ValueError):
                print(f"{name}: FAILED\n{out.stderr[-2000:]}", flush=True)
    else:
        runs = [("one launch", "default", {"XVR_DRR_FWD_SLABS": "0"}), ("one launch, natural layout", "natural", {"XVR_DRR_FWD_SLABS": "0"}),
                ("one launch, 116 x one pose", "samepose", {"XVR_DRR_FWD_SLABS": "0"}),
                ("one launch, no jacobian", "nojac", {"XVR_DRR_FWD_SLABS": "0"})]
        for axis in (2, 0, 1):
            for n in (4, 8, 12):
                runs.append((f"{n} slabs along axis {axis}", "default", {"XVR_DRR_FWD_SLABS": str(n), "XVR_DRR_FWD_SLAB_AXIS": str(axis)}))
        runs += [("6 slabs along axis 2", "default", {"XVR_DRR_FWD_SLABS": "6", "XVR_DRR_FWD_SLAB_AXIS": "2"}),
                 ("16 slabs along axis 2", "default", {"XVR_DRR_FWD_SLABS": "16", "XVR_DRR_FWD_SLAB_AXIS": "2"}),
                 ("natural layout, 4 slabs along axis 2", "natural", {"XVR_DRR_FWD_SLABS": "4", "XVR_DRR_FWD_SLAB_AXIS": "2"}),
                 ("natural layout, 8 slabs along axis 2", "natural", {"XVR_DRR_FWD_SLABS": "8", "XVR_DRR_FWD_SLAB_AXIS": "2"}),
                 ("no jacobian, 8 slabs along axis 2", "nojac", {"XVR_DRR_FWD_SLABS": "8", "XVR_DRR_FWD_SLAB_AXIS": "2"}),
                 ("116 x one pose, 8 slabs along axis 2", "samepose", {"XVR_DRR_FWD_SLABS": "8", "XVR_DRR_FWD_SLAB_AXIS": "2"}),
                 ("default options", "default", {})]
        ref = None
        for name, mode, env in runs:
            out = subprocess.run([sys.executable, __file__, "one", mode], env=dict(os.environ, **env), capture_output=True, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
            except (IndexError, ValueError):
                print(f"{name}: FAILED\n{out.stderr[-2000:]}", flush=True)
                continue
            if name == "one launch":
                ref = d
            note = ""
            if mode == "default" and ref is not None:
                note = "  bits identical" if d["sha"] == ref["sha"] else f"  image sum differs by {abs(d['sum'] - ref['sum']) / abs(ref['sum']):.1e} (relative)"
            print(f"{name:40s} forward {d['ms']:.3f} ms (min {d['min_ms']:.3f}){note}", flush=True)

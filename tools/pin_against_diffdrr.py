#!/usr/bin/env python
"""Pin the oracle (and through it the HIP kernels) against the REAL diffdrr -- one command, on any machine where
``import diffdrr`` works (``pip install diffdrr==0.6.0``, the version xvr pins: /root/reference/uv.lock:955-977).

    python tools/pin_against_diffdrr.py            # writes tests/golden/diffdrr_pin.npz, prints the matching knobs

It cannot run in the build container (the package is absent and there is no network); it is the ready-to-run recipe
VERDICT round 1 asked for.  What it does:

1. renders committed, seeded inputs with the real ``diffdrr.renderers.Siddon`` / ``Trilinear`` modules, called
   exactly as xvr calls them (``renderer(volume, source, target, img, mask=...)``,
   /root/reference/src/xvr/model/trainer.py:288) -- the `make_case` inputs of tests/conftest.py (seed 11 and 12)
   and one 128 x 128 DeepFluoro-geometry pose (C1) -- forward, and the gradients w.r.t. source / target / volume;
2. when the package's ``DRR`` module can be built from tensors (needs torchio), renders the C1 pose through
   ``DRR(subject, sdd, height, delx, ...)(pose)`` as /root/reference/src/xvr/renderer/load.py:32-44 constructs it;
3. grid-searches the oracle's ``RenderSpec`` knobs (A2 step normalisation, A4 voxel_shift, A5
   dims offset, A6 align_corners, clip_to_volume, per_ray_clamp) for the set that reproduces every vector, prints
   the table of errors, and stores inputs, outputs and the winning knobs in ``tests/golden/diffdrr_pin.npz``.

``tests/test_diffdrr_pin.py`` consumes that file when present: the oracle (CPU) and the HIP kernels (``-m gpu``)
must reproduce the real package's outputs with the recorded knobs.  The day the file exists, parity is pinned without
touching a kernel; if the winning knobs differ from ``xvr_amd.spec.RenderSpec``'s defaults, change the defaults.
"""
import argparse
import dataclasses
import inspect
import itertools
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

OUT = ROOT / "tests" / "golden" / "diffdrr_pin.npz"


def _accepts(fn, **kw):
    """Only the keyword arguments the installed version's signature takes (the pin must survive minor API drift)."""
    try:
        names = set(inspect.signature(fn).parameters)
    except (TypeError, ValueError):
        return kw
    return {k: v for k, v in kw.items() if k in names}


def real_render(renderer, volume, source, target, img, voxel_shift, n_points, mask=None, grads=False):
    """One call of the real renderer module; returns dict(out[, gvol, gsrc, gtgt])."""
    import diffdrr.renderers as R

    cls = R.Trilinear if renderer == "trilinear" else R.Siddon
    mod = cls(**_accepts(cls.__init__, voxel_shift=voxel_shift))
    v, s, t = volume.clone(), source.clone(), target.clone()
    if grads:
        for x in (v, s, t):
            x.requires_grad_(True)
    kw = dict(mask=mask)
    if renderer == "trilinear":
        kw["n_points"] = n_points
    out = mod(v, s, t, img, **_accepts(mod.forward, **kw))
    res = {"out": out.detach()}
    if grads:
        w = torch.rand(out.shape, generator=torch.Generator().manual_seed(4))
        (out * w).sum().backward()
        res.update(w=w, gvol=v.grad, gsrc=s.grad, gtgt=t.grad)
    return res


def c1_case(quick=False):
    """One pose at C1's geometry (scripts/deepfluoro/train/de_novo.sh:24-32) over a small seeded phantom.  ``quick``: a quarter
    of the detector and of the volume per axis -- the rehearsal of tests/test_diffdrr_pin.py, not the pin."""
    from oracle.diffdrr_restated import _apply, rays_from_pose
    from xvr_amd.data import make_phantom
    from xvr_amd.pose import convert

    if quick:
        vol, lab = make_phantom((22, 20, 24), n_ellipsoids=6, n_labels=4, seed=7)
        affine = torch.diag(torch.tensor([12.8, 14.4, 12.0, 1.0]))
        affine[:3, 3] = -(affine[:3, :3] @ ((torch.tensor(vol.shape, dtype=torch.float32) - 1) / 2))
        pose = convert(torch.tensor([[175.0, 12.0, -6.0]]), torch.tensor([[20.0, 760.0, -35.0]]), parameterization="euler_angles",
                       convention="ZXY", degrees=True)
        src, tgt = rays_from_pose(pose.matrix, 32, 32, 1020.0, 2.1764375 * 4, 2.1764375 * 4, 0.0, 0.0, "AP", True)
        img = (tgt - src).norm(dim=-1).unsqueeze(1)
        affinv = torch.linalg.inv(affine)[None]
        return dict(volume=vol, mask=lab, source=_apply(affinv, src), target=_apply(affinv, tgt), img=img, affine=affine,
                    pose=pose.matrix, height=32, width=32, sdd=1020.0, delx=2.1764375 * 4)
    vol, lab = make_phantom((88, 80, 96), n_ellipsoids=12, n_labels=4, seed=7)
    affine = torch.diag(torch.tensor([3.2, 3.6, 3.0, 1.0]))
    affine[:3, 3] = -(affine[:3, :3] @ ((torch.tensor(vol.shape, dtype=torch.float32) - 1) / 2))
    pose = convert(torch.tensor([[175.0, 12.0, -6.0]]), torch.tensor([[20.0, 760.0, -35.0]]), parameterization="euler_angles",
                   convention="ZXY", degrees=True)
    src, tgt = rays_from_pose(pose.matrix, 128, 128, 1020.0, 2.1764375, 2.1764375, 0.0, 0.0, "AP", True)
    img = (tgt - src).norm(dim=-1).unsqueeze(1)
    affinv = torch.linalg.inv(affine)[None]
    return dict(volume=vol, mask=lab, source=_apply(affinv, src), target=_apply(affinv, tgt), img=img, affine=affine,
                pose=pose.matrix, height=128, width=128, sdd=1020.0, delx=2.1764375)


def real_drr_module(case, renderer, voxel_shift):
    """DRR(subject, ...)(pose) of the real package on the C1 case; None when it cannot be built from tensors."""
    try:
        import torchio
        from diffdrr.data import read
        from diffdrr.drr import DRR
        from diffdrr.pose import RigidTransform

        image = torchio.ScalarImage(tensor=case["volume"][None], affine=case["affine"].numpy())
        subject = read(image, **_accepts(read, orientation="AP", center_volume=False))
        drr = DRR(subject, case["sdd"], case["height"], case["delx"],
                  **_accepts(DRR.__init__, renderer=renderer, reverse_x_axis=True, voxel_shift=voxel_shift))
        with torch.no_grad():
            return drr(RigidTransform(case["pose"])).detach()
    except Exception as e:   # noqa: BLE001 -- the renderer-level vectors above are the pin; this one is a bonus
        print(f"  (DRR module vector skipped: {type(e).__name__}: {e})")
        return None


def knob_grid(renderer):
    # (eps_in_xyz moves the sample points by <= 1e-8 voxels: it can only ever be told apart in float64 vectors, and one kernel
    #  serves both; filter_intersections_outside_volume is numerically neutral under per_ray_clamp and is listed so that the
    #  report names the upstream setting)
    common = dict(norm_dims_offset=[0, +1, -1], align_corners=[False, True], eps_in_xyz=[True, False])
    if renderer == "trilinear":
        grid = dict(common, step_mode=["n_points", "n_minus_1"], clip_to_volume=[False, True, "batch"])
    else:
        grid = dict(common, per_ray_clamp=[True, False], filter_intersections_outside_volume=[True, False])
    keys = list(grid)
    for values in itertools.product(*(grid[k] for k in keys)):
        yield dict(zip(keys, values))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(OUT))
    ap.add_argument("--quick", action="store_true", help="small C1 stand-in (the rehearsal against a planted renderer in the test suite)")
    args = ap.parse_args(argv)
    try:
        import diffdrr
    except ImportError:
        raise SystemExit("diffdrr is not importable here: run this where `pip install diffdrr==0.6.0` has been done")
    if "xvr_amd" in (getattr(diffdrr, "__doc__", "") or ""):
        raise SystemExit("`diffdrr` resolves to xvr_amd.compat's alias, not to the real package")
    print(f"diffdrr {getattr(diffdrr, '__version__', '?')} from {Path(diffdrr.__file__).parent}")
    from conftest import make_case
    from oracle.diffdrr_restated import RenderSpec, drr_from_pose, render as oracle_render

    store, report = {}, []
    cases = {"case11": make_case(seed=11), "case12": make_case(seed=12), "c1": c1_case(args.quick)}
    for renderer in ("trilinear", "siddon"):
        for shift in (0.5, 0.0):
            n_points = 60
            best = None
            real = {}
            for name, case in cases.items():
                real[name] = real_render(renderer, case["volume"], case["source"], case["target"], case["img"], shift, n_points, grads=True)
                real[name + "_mask"] = real_render(renderer, case["volume"], case["source"], case["target"], case["img"], shift, n_points,
                                                   mask=case["mask"])
            for knobs in knob_grid(renderer):
                spec = RenderSpec(renderer=renderer, voxel_shift=shift, n_points=n_points, **knobs)
                worst = 0.0
                for name, case in cases.items():
                    v, s, t = (case[k].clone().requires_grad_(True) for k in ("volume", "source", "target"))
                    try:
                        out = oracle_render(v, s, t, case["img"], spec)
                        (out * real[name]["w"]).sum().backward()
                        outm = oracle_render(case["volume"], case["source"], case["target"], case["img"], spec, case["mask"])
                    except Exception:   # noqa: BLE001
                        worst = float("inf")
                        break
                    for got, want in ((out, real[name]["out"]), (v.grad, real[name]["gvol"]), (s.grad, real[name]["gsrc"]),
                                      (t.grad, real[name]["gtgt"]), (outm, real[name + "_mask"]["out"])):
                        if got.shape != want.shape:
                            worst = float("inf")
                            break
                        worst = max(worst, ((got - want).abs().max() / want.abs().max().clamp_min(1e-12)).item())
                report.append((renderer, shift, knobs, worst))
                if best is None or worst < best[1]:
                    best = (knobs, worst)
            tag = f"{renderer}_shift{shift}"
            print(f"{tag}: best knobs {best[0]}  worst relative error {best[1]:.3e}")
            store[tag + "_knobs"] = json.dumps(best[0])
            store[tag + "_err"] = best[1]
            for name in real:
                for k, val in real[name].items():
                    store[f"{tag}_{name}_{k}"] = val.numpy()
            img = real_drr_module(cases["c1"], renderer, shift)
            if img is not None:
                store[tag + "_c1_drr_module"] = img.numpy()
                spec = RenderSpec(renderer=renderer, voxel_shift=shift, **best[0])
                c = cases["c1"]
                mine = drr_from_pose(c["volume"], c["affine"], c["pose"], c["height"], c["width"], c["sdd"], c["delx"], c["delx"], 0.0, 0.0, spec,
                                     orientation="AP", reverse_x_axis=True)
                err = ((mine - img).abs().max() / img.abs().max()).item()
                print(f"  DRR module (n_points = 500 default) vs oracle with the winning knobs: {err:.3e}")
                store[tag + "_c1_drr_module_err"] = err
    for name, case in cases.items():   # the inputs travel with the outputs: the GPU box has neither diffdrr nor this script's RNG
        for k in ("volume", "mask", "source", "target", "img"):
            store[f"in_{name}_{k}"] = case[k].numpy()
    store["diffdrr_version"] = str(getattr(diffdrr, "__version__", "?"))
    np.savez_compressed(args.out, **store)
    print(f"wrote {args.out}")
    print("all candidates:")
    for renderer, shift, knobs, worst in sorted(report, key=lambda r: (r[0], r[1], r[3])):
        print(f"  {renderer:9s} shift {shift}: {worst:.3e}  {knobs}")
    bad = [k for k in store if k.endswith("_err") and not k.endswith("module_err") and store[k] > 1e-4]
    if bad:
        print("NOT PINNED: no knob set reproduces", bad, "-- the restatement itself needs a change (see the table above)")
        return 1
    defaults = {f.name: f.default for f in dataclasses.fields(RenderSpec)}
    print("PINNED.  Defaults that differ from the winning knobs:")
    for tag in (k[:-6] for k in store if k.endswith("_knobs")):
        for k, v in json.loads(store[tag + "_knobs"]).items():
            if defaults.get(k) != v:
                print(f"  {tag}: {k} = {v} (default {defaults.get(k)})")
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Shared measuring code of bench.py's secondary legs and of the stand-alone tools (tools/bench_register.py,
tools/bench_training_step.py): the same functions produce the driver's JSON line and the committed profiles/r04_* files.

  c4_register    BASELINE.json configs[3]: one iteration of the multiscale gradient-NCC pose refinement (render -> transforms ->
                 mNCC + gradNCC -> backward -> Adam -> plateau scheduler, /root/reference/src/xvr/registrar/base.py:245-280) on a
                 512^3 CT at the two pyramid levels of a 2048^2 X-ray with scales "8,4" (256^2, then 512^2), one start and 8 starts
                 advanced as one batch
  c5_train_step  BASELINE.json configs[4], one GPU's share: the render side of one training step
                 (/root/reference/src/xvr/model/trainer.py:185-230) -- HU -> density, two 8-channel renders of 116 DRRs, the
                 stand-in for the regressor, transforms + PoseRegressionLoss, backward
"""
import time

import torch


PROFILES = __import__("pathlib").Path(__file__).resolve().parents[1] / "profiles"


def parse_pmc_summary(path):
    """{kernel name (as printed, truncated to 100 characters): {counter: PER-DISPATCH value, "dispatches": n}} of one
    profiles/rNN_*_rocprof_summary.md (tools/summarize_profile.py prints totals over the dispatches of a kernel)."""
    import re

    out, cur = {}, None
    for line in open(path):
        m = re.match(r"### `(.*)`\s+\((\d+) dispatches\)", line)
        if m:
            cur = out.setdefault(m.group(1), {"dispatches": int(m.group(2))})
            continue
        m = re.match(r"- ([A-Za-z0-9_]+): ([-+0-9.eE]+)\s*$", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2)) / cur["dispatches"]
        elif line.startswith("## "):
            cur = None
    return out


def committed_counters(kind, pattern, profiles=None):
    """The per-dispatch PMC counters of the kernel instantiation matching ``pattern`` (a regex) in the NEWEST committed
    profiles/rNN_<kind>_rocprof_summary.md, with the unit count (volume-touching samples / voxel segments) of the bench run they
    were collected under (profiles/rNN_<kind>_bench_under_trace.json): {counter: value, "units": n, "file": ..., "units_file": ...};
    None when nothing is committed for it."""
    import json
    import re

    root = __import__("pathlib").Path(profiles) if profiles else PROFILES
    for md in sorted(root.glob(f"r[0-9][0-9]_{kind}_rocprof_summary.md"), reverse=True):
        trace = md.with_name(md.name.replace("_rocprof_summary.md", "_bench_under_trace.json"))
        if not trace.exists():
            continue
        try:
            units = json.loads(trace.read_text())["roofline"]["units_per_launch"]
        except (ValueError, KeyError):
            continue
        hits = [v for k, v in parse_pmc_summary(md).items() if re.search(pattern, k)]
        if hits and units:
            best = max(hits, key=lambda v: len(v))   # (the instantiation every pass saw)
            return {**best, "units": float(units), "file": f"profiles/{md.name}", "units_file": f"profiles/{trace.name}"}
    return None


def _kernel_table(events):
    per = {}
    for name, e0, e1 in events:
        per.setdefault(name, []).append(e0.elapsed_time(e1))
    return {k: {"launches": len(v), "avg_ms": sum(v) / len(v)} for k, v in per.items()}


def c4_register(dev, subject, sizes=((256, 0.1360 * 8), (512, 0.1360 * 4)), n_it=120, starts=8, extra=()):
    """-> {"256": {"single": ms per iteration, "batched8": ms per pose-iteration, "kernels": {...}}, "512": {...}}.
    `extra`: further Registrar keyword sets timed as single runs (sigma, equalize, parameterization)."""
    from xvr_amd import renderers
    from xvr_amd.drr import DRR
    from xvr_amd.pose import convert
    from xvr_amd.registrar import Registrar

    out = {}
    for H, delx in sizes:
        drr = DRR(subject, 1020.0, H, delx, renderer="trilinear", reverse_x_axis=False, voxel_shift=0.0).to(dev)
        rot, xyz = torch.tensor([[3.1, 0.05, -0.02]], device=dev), torch.tensor([[5.0, 750.0, -8.0]], device=dev)
        with torch.no_grad():
            gt = drr(convert(rot + 0.03, xyz + 5.0, parameterization="euler_angles", convention="ZXY"))
        init = convert(rot.cpu(), xyz.cpu(), parameterization="euler_angles", convention="ZXY")

        def steady(reg_kw, batch=None, repeats=3):
            """-> (median over `repeats` runs of [the median of a run's per-check-block means over its last 6 blocks], (min, median,
            max) over the runs, the last run's result).  One un-repeated mean of 50 per-iteration times let a single 27 ms host
            hiccup move a 0.36 ms figure by 2.5 x; a block is `check_every` = 8 iterations between two host
            syncs, timed as one."""
            import statistics

            per_run, r0 = [], None
            for _ in range(repeats):
                R = Registrar(drr, scales="1", n_itrs=str(n_it), max_n_plateaus=100, **reg_kw)
                torch.cuda.synchronize()
                res = R.run(gt, init) if batch is None else R.run_batch(gt, batch)
                torch.cuda.synchronize()
                r0 = res if batch is None else res[0]
                t = r0["times"][1:]                       # ([0] is the reference's leading 0.0)
                blocks = [sum(t[i:i + 8]) / len(t[i:i + 8]) for i in range(0, len(t), 8)][-6:]
                per_run.append(statistics.median(blocks) * 1e3)
            return statistics.median(per_run), [min(per_run), statistics.median(per_run), max(per_run)], r0

        entry, spread = {}, {}
        ms, spread["single"], r0 = steady({})
        entry["single"] = ms
        entry["ncc"] = [r0["nccs"][0], r0["nccs"][-1]]
        g = torch.Generator().manual_seed(0)
        inits = convert(rot.cpu() + (torch.rand(starts, 3, generator=g) - 0.5) * 0.06, xyz.cpu() + (torch.rand(starts, 3, generator=g) - 0.5) * 10.0,
                        parameterization="euler_angles", convention="ZXY")
        ms, sp, _ = steady({}, inits)
        entry[f"batched{starts}"] = ms / starts
        spread[f"batched{starts}"] = [x / starts for x in sp]
        for kw in extra:
            key = "single " + ", ".join(f"{k}={v}" for k, v in kw.items())
            entry[key], spread[key], r0 = steady(dict(kw))
        entry["min_median_max_of_3_runs"] = spread
        # the kernels of one iteration, eagerly (HIP events do not live inside a replayed graph)
        renderers.PROFILER = []
        Registrar(drr, scales="1", n_itrs="12", max_n_plateaus=100, use_graph=False).run(gt, init)
        torch.cuda.synchronize()
        ev, renderers.PROFILER = renderers.PROFILER, None
        entry["kernels"] = _kernel_table(ev)
        out[str(H)] = entry
        del drr
    return out


def c5_train_step(dev, size=512, B=116, H=256, n=9, warm=3):
    """-> {"ms_per_step": wall ms, "phases": {name: ms by HIP events}, "kernels": {...}}"""
    from xvr_amd import renderers
    from xvr_amd.data import make_phantom, read, transform_hu_to_density
    from xvr_amd.drr import DRR
    from xvr_amd.loss import PoseRegressionLoss
    from xvr_amd.metrics import XrayTransforms
    from xvr_amd.pose import convert
    from xvr_amd.training import get_random_pose, render_samples

    vol, lab = make_phantom(size, n_ellipsoids=64, n_labels=8, seed=0, device=dev)
    hu = vol * 1400 - 1000
    del vol
    drr = DRR(read(hu, lab, spacing=(512.0 / size,) * 3, orientation="AP", hu=True), 1020.0, H, 1.08821875 * 256 / H, renderer="trilinear",
              reverse_x_axis=False).to(dev)
    drr.register_buffer("volume", hu)
    transforms = XrayTransforms(H)
    lossfn = PoseRegressionLoss(1020.0).to(dev)
    g = torch.Generator().manual_seed(0)
    marks = {}

    def lap(name, e_prev):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.setdefault(name, []).append((e_prev, e))
        return e

    def step():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        pose = get_random_pose(135.0, 225.0, -45.0, 45.0, -15.0, 15.0, -150.0, 150.0, 450.0, 1000.0, -150.0, 150.0, B, generator=g).to(dev)
        e = lap("sample poses", e)
        # (lazy: the density volume is never written -- the map rides in the renders' packing pass, xvr_amd.data.HUDensity)
        tmp = transform_hu_to_density(drr.volume, float(torch.empty(1).uniform_(1.0, 10.0, generator=g)), lazy=True)
        e = lap("HU -> density", e)
        with torch.no_grad():
            img, mask, keep = render_samples(drr, tmp, drr.mask, drr.affine_inverse, pose)
        e = lap("render #1 (no grad, 8 channels)", e)
        rot, xyz = pose.convert("quaternion_adjugate")
        rot = (rot + 0.01 * torch.randn(rot.shape, generator=g).to(dev)).requires_grad_()
        xyz = (xyz + 5.0 * torch.randn(xyz.shape, generator=g).to(dev)).requires_grad_()
        pred_pose = convert(rot, xyz, parameterization="quaternion_adjugate")
        e = lap("stand-in for the regressor", e)
        pred_img, pred_mask, _ = render_samples(drr, tmp, drr.mask, drr.affine_inverse, pred_pose)
        e = lap("render #2 (grad)", e)
        loss, *_ = lossfn(transforms(img), mask, pose, transforms(pred_img), pred_mask, pred_pose)
        e = lap("transforms + PoseRegressionLoss", e)
        loss.mean().backward()
        lap("backward (loss + render)", e)
        return loss

    for _ in range(warm):
        step()
    marks.clear()
    renderers.PROFILER = []
    import statistics

    walls = []
    for _ in range(n):     # every step timed on its own (host clock around a synchronised step): median, not one mean of n
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) * 1e3)
    total = statistics.median(walls)
    ev, renderers.PROFILER = renderers.PROFILER, None
    return {"ms_per_step": total, "min_median_max_ms": [min(walls), total, max(walls)], "steps": n,
            "config": f"{size}^3 -> {H}^2, batch {B}, 8 label channels, render side only (no regressor)",
            "phases": {k: statistics.median(a.elapsed_time(b) for a, b in v) for k, v in marks.items()}, "kernels": _kernel_table(ev)}

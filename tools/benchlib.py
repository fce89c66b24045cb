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



# ---- what binds the render kernels: floors priced from the committed microbenchmarks and PMC summaries (moved out of bench.py in
# round 6: the driver's line carries only the dominant kernel's {unit, floor_ms, frac}; the tables go to bench_full.json) ----
import json
import sys

ROOT = PROFILES.parent
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured copy peak
HBM_COPY_GBS = 6290.0
# What actually binds the render kernels (their taps are served by L1 / L2 / LDS, not by HBM): unit costs measured on this
# chip by the committed microbenchmarks, applied to the live unit count.  floor_ms = the time the launch would take if the
# binding unit were busy every clock and nothing else cost anything.
CUS, CLK_GHZ = 256, 2.4
VALU_CLK = 3.0   # clocks a plain wave64 vector instruction occupies its SIMD (2.9-3.4 measured with 4 wavefronts per SIMD,
                 # tools/microbench/valu_issue.hip, profiles/r03_microbench_valu_issue.txt; v_pk_*_f32 7.5, v_mad_u64_u32 6.0)
# candidate units per timed call: (unit, wavefront instructions per 64 units of work, clocks each on the unit's 256 CU-wide or
# 1024 SIMD-wide resource, source).  The instruction and line counts are NOT literals: they are read at run time from the newest
# committed PMC summaries (profiles/rNN_{trilinear,siddon}_rocprof_summary.md, written by tools/profile.sh + summarize_profile.py)
# and divided by the unit count of the very run they were collected under (profiles/rNN_*_bench_under_trace.json), so a kernel
# edit followed by a profile run re-prices the floors without touching this file (tests/test_bench_contract.py checks the
# parse against the files).
# fabric_bandwidth: an L2 miss moves one whole 128-byte line (tools/microbench/fetch_calib.hip, profiles/r04_fetch_calibration.txt);
# random lines of a working set far beyond the Infinity Cache arrive at 43-46 G lines/s (5.5-5.9 TB/s) -- 0.054 clocks per line for
# the chip.
LINE_CLK = CLK_GHZ / 44.5
# timed call -> (which summary, regex of the kernel instantiation that IS the call's steady state)
BINDING_KERNELS = {
    "trilinear_backward": ("trilinear", r"k_trilinear_splat_b16"),
    "trilinear_forward+jac": ("trilinear", r"k_trilinear_fwd<true, 0, false, [1-9]"),
    "trilinear_forward": ("trilinear", r"k_trilinear_fwd<false, 0, false, [1-9]"),
    "siddon_backward": ("siddon", r"k_siddon_gather_vol2"),
    "siddon_forward+jac": ("siddon", r"k_siddon_slab<true, true(, false)?>"),
    "siddon_forward": ("siddon", r"k_siddon_slab<false, true(, false)?>"),
    # the recalled index map (dims = shape + 1): the slab march's NX instantiations and the ray-driven brick splat
    "siddon_backward@nx": ("siddon_nx", r"k_siddon_splat"),
    "siddon_forward+jac@nx": ("siddon_nx", r"k_siddon_slab<true, true, true>"),
    "siddon_forward@nx": ("siddon_nx", r"k_siddon_slab<false, true, true>"),
}
_BINDING_CACHE = {}


def binding_candidates(base):
    """[(unit, wavefront instructions per 64 units, clocks each, width, source)] for one timed call, from the committed counters."""
    if base in _BINDING_CACHE:
        return _BINDING_CACHE[base]
    cands = []
    kind, pattern = BINDING_KERNELS.get(base, (None, None))
    c = committed_counters(kind, pattern) if kind else None
    if c:
        per64 = lambda counter: c[counter] / (c["units"] / 64.0)   # noqa: E731
        src = f"{c['file']} / units of {c['units_file']}"
        if base == "trilinear_backward" and "SQ_THREAD_CYCLES_VALU" in c:
            # tools/microbench/lds_atomics.hip: a ds_add_u32 wavefront instruction costs >= 4.4 LDS clocks however few lanes are
            # live; 8 per sample; live lanes per vector instruction from the same PMC pass
            live = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"]
            cands.append(("lds_atomic_issue", 8 / (live / 64.0), 4.4, CUS, f"profiles/r02_microbench_lds_atomics.txt; {live:.1f} live lanes ({src})"))
        if base.startswith("trilinear_forward"):
            # tools/microbench/gather.hip: a 64-lane gather costs ~15 clk per CU on one 128-B line and 10-14 more per further line;
            # two 16-byte gathers per sample (y-pair copy), ~3 lines by that estimate
            cands.append(("texture_address", 2, 36.0, CUS, "profiles/r01_microbench_gather_lines.txt"))
        if base.startswith("siddon_forward+jac") and "TA_BUSY_avr" in c:
            # clocks the CUs' texture-address units were busy (TA_BUSY_avr: mean over the TA instances), per 64 voxel segments
            cands.append(("texture_address", 1.0, c["TA_BUSY_avr"] / (c["units"] / 64.0) * CUS, CUS, f"TA_BUSY_avr ({src})"))
        if "SQ_INSTS_VALU" in c:
            cands.append(("valu_issue", per64("SQ_INSTS_VALU"), VALU_CLK, 4 * CUS, f"profiles/r03_microbench_valu_issue.txt x SQ_INSTS_VALU ({src})"))
        if "TCC_EA0_RDREQ_sum" in c:
            cands.append(("fabric_bandwidth", per64("TCC_EA0_RDREQ_sum"), LINE_CLK, 1, f"profiles/r04_fetch_calibration.txt x TCC_EA0_RDREQ ({src})"))
    _BINDING_CACHE[base] = cands
    return cands


def binding_floor(tag, units, avg_ms, variant=""):
    """The unit with the largest floor for one timed call, and every candidate's floor next to it.  ``variant`` "nx": the leg renders
    under a non-exact Siddon index map (other kernels, other committed profile)."""
    cands = binding_candidates(tag.split("[")[0] + ("@" + variant if variant else ""))
    if not cands or not units:
        return None
    floors = {}
    for unit, per64, clk, width, source in cands:
        floors[unit] = {"floor_ms": units / 64.0 * per64 * clk / width / (CLK_GHZ * 1e9) * 1e3, "clk_per_wave_instr": clk, "source": source}
    unit = max(floors, key=lambda u: floors[u]["floor_ms"])
    return {"unit": unit, "floor_ms": floors[unit]["floor_ms"], "frac": floors[unit]["floor_ms"] / avg_ms,
            "clk_per_wave_instr": floors[unit]["clk_per_wave_instr"], "source": floors[unit]["source"],
            "floors_ms": {u: v["floor_ms"] for u, v in floors.items()}}


# which profiled kernels make up each timed C-ABI call (one call may launch several kernels)
TRAFFIC_KERNELS = {
    "trilinear_forward": ["k_trilinear_fwd"],
    "trilinear_backward": ["k_trilinear_splat_b16", "k_trilinear_gather_tab", "k_gather_prep", "k_gather_cull", "k_trilinear_bwd"],
    "siddon_forward": ["k_siddon<", "k_siddon_slab"],
    "siddon_backward": ["k_siddon_gather_vol", "k_gather_prep", "k_gather_cull", "k_siddon<"],
    "backward_from_jac": ["k_backward_from_jac"],
}


def pmc_traffic(tag, variant=""):
    """HBM-side bytes per launch of the kernels behind one timed call, from the committed rocprofv3 PMC
    passes (profiles/traffic.json, written by tools/summarize_profile.py: FETCH_SIZE + WRITE_SIZE, in
    bytes MOVED: the fetch side is the reported FETCH_SIZE doubled, as calibrated in profiles/r04_fetch_calibration.txt).
    None when not profiled."""
    path = ROOT / "profiles" / "traffic.json"
    if not path.exists():
        return None
    try:
        table = json.loads(path.read_text())
    except ValueError:
        return None
    base = tag.split("[")[0].split("+")[0]
    keys = TRAFFIC_KERNELS.get(base)
    if not keys:
        return None
    nx = variant == "nx"
    if base == "siddon_backward":   # (the exact map's voxel gather, or the brick splat of a non-exact one: never both)
        keys = [k for k in keys if k != "k_siddon_gather_vol"] + ["k_siddon_splat"] if nx else keys
    fam = {}   # kernel family (name up to its template list) -> traffic of each profiled instantiation that matches
    for name, v in table.items():
        if "k_siddon<" in name:   # k_siddon<MODE, ...>: 0 forward, 1 forward + jacobian, 2 backward
            mode = "2" if base == "siddon_backward" else ("1" if "+jac" in tag else "0")
            if f"k_siddon<{mode}," not in name:
                continue
        if "k_siddon_slab<" in name and (("k_siddon_slab<true" in name) != ("+jac" in tag) or base != "siddon_forward"
                                         or name.split("k_siddon_slab<")[1].split(">")[0].endswith(", true") != nx):
            continue
        if any(k in name for k in keys) and (("fwd<true" in name) == ("+jac" in tag) or "fwd<" not in name):
            fam.setdefault(name.split("<")[0], []).append(v.get("fetch_bytes", 0.0) + v.get("write_bytes", 0.0))
    # different kernels of one call add up; instantiations of one kernel (volume layouts) are alternatives: their mean
    return sum(sum(v) / len(v) for v in fam.values()) if fam else None


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


def c5_train_step(dev, size=512, B=116, H=256, n=9, warm=3, drr_kwargs=None):
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
              reverse_x_axis=False, **(drr_kwargs or {})).to(dev)
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
            "config": f"{size}^3 -> {H}^2, batch {B}, 8 label channels, render side only (no regressor)" + (f", spec {drr_kwargs}" if drr_kwargs else ""),
            "phases": {k: statistics.median(a.elapsed_time(b) for a, b in v) for k, v in marks.items()}, "kernels": _kernel_table(ev)}

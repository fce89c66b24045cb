#!/usr/bin/env python
"""Benchmark of the DRR render path on MI355X: DRRs/s, forward + backward (pose and voxel gradient).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: render B poses of a 512^3 volume onto a 256^2
detector through DRR.forward (trilinear, n_points=500) and backpropagate a weighted sum to the pose
parameters AND to the voxels.  Workload = BASELINE.json configs[1] ("Single 512^3 CT, trilinear
fwd+bwd, 256x256 detector, batch_size=116 on 1xMI355X"); synthetic seeded phantom and poses
(SURVEY.md section 8d).  With N > 1 each rank renders its own B poses of a replicated volume (weak
scaling; the unit -- a pose -- is independent) and the rendered DRRs are all-gathered over RCCL.

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel, timed live with HIP events
on the launch stream; `cpu_baseline` is the oracle (the torch-ops restatement of the reference's CPU
render path) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import benchlib  # noqa: E402  (secondary legs, and the floors priced from the committed counters: details only, bench_full.json)
from benchlib import binding_floor, pmc_traffic  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured copy peak
HBM_COPY_GBS = 6290.0
TAP_BYTES_PER_SAMPLE = 32   # 8 taps x 4 B; the voxel backward accumulates into the same 8

def deepfluoro_poses(batch, seed):
    from xvr_amd.training import get_random_pose

    g = torch.Generator().manual_seed(seed)
    # ranges of scripts/deepfluoro/train/de_novo.sh:24-29
    return get_random_pose(135.0, 225.0, -45.0, 45.0, -15.0, 15.0, -150.0, 150.0, 450.0, 1000.0, -150.0, 150.0,
                           batch, generator=g)


class Exchange:
    """The exchange step of the path for N > 1 (and --force-dist): every rank ends up with every rendered DRR -- one
    all_gather_into_tensor over RCCL, issued async right after the forward and waited for after the backward.  The unequal
    shares of a ragged strong-scaling split are padded to the largest share (no backend gathers uneven tensors in one call)."""

    def __init__(self, world, B, Bmax, H, dev, volume_grad=True, buckets=8, slabs=4):
        self.B = B
        self.gathered = torch.empty(world * Bmax, 1, H, H, device=dev)
        self.send = torch.zeros(Bmax, 1, H, H, device=dev) if Bmax != B else None
        self.handle = None
        # the second exchange of the fwd+bwd(pose+voxel) step: every rank rendered different poses of the SAME
        # volume, so the voxel gradients are summed.  slabs > 1 (default): the backward computes the voxel gradient in x slabs and
        # every finished slab goes into an async all-reduce while the next one is computed (xvr_amd.distributed.SlabAllReduce) --
        # only the last slab's collective is exposed.  slabs <= 1: bucketed async all-reduces issued when the backward has written
        # the whole gradient (allreduce_volume_grad_bucketed).  Either way waited for at the end of the step.
        from xvr_amd.distributed import SlabAllReduce
        self.volume_grad, self.buckets, self.works = volume_grad, buckets, []
        self.slabs = int(slabs)
        self.slab = SlabAllReduce(self.slabs, force=True) if self.slabs > 1 else None
        self.pending = None

    def post_forward(self, img):
        if self.send is not None:
            self.send[:self.B].copy_(img.detach())
        self.handle = dist.all_gather_into_tensor(self.gathered, img.detach() if self.send is None else self.send, async_op=True)

    def pre_backward(self, leaf=None):
        if self.slab is not None:
            if self.volume_grad:
                self.slab.install(leaf)
            else:
                self.slab.remove()

    def post_backward(self, grad):
        if self.slab is not None:
            self.slab.remove()
        if self.volume_grad and grad is not None:
            if self.slab is not None and self.slab.fired():
                self.pending = grad
            else:
                from xvr_amd.distributed import allreduce_volume_grad_bucketed
                self.works = allreduce_volume_grad_bucketed(grad, self.buckets, force=True)

    def wait(self):
        if self.handle is not None:
            self.handle.wait()
            self.handle = None
        for w in self.works:
            w.wait()
        self.works = []
        if self.pending is not None:
            self.slab.finish(self.pending)
            self.pending = None


def render_leg(dev, subject, renderer, voxel_grad, rot0, xyz0, H, delx, n_points, steps, warmup, exchange=None, update_volume=False,
               drr_kwargs=None, B_total=None):
    """One benchmark leg: `steps` timed passes of DRR.forward of the poses + backward of a weighted sum to (rot, xyz) and, with
    voxel_grad, to the voxels.  Returns the timings, the per-kernel HIP-event table and the roofline object of the leg."""
    from xvr_amd import renderers
    from xvr_amd.drr import DRR
    from xvr_amd.pose import convert

    B = rot0.shape[0]
    kw = {"n_points": n_points} if renderer == "trilinear" else {}
    drr = DRR(subject, 1020.0, H, delx, renderer=renderer, reverse_x_axis=False, **(drr_kwargs or {})).to(dev)
    density = drr.density.clone().requires_grad_(voxel_grad)
    rot = rot0.detach().clone().to(dev).requires_grad_(True)
    xyz = xyz0.detach().clone().to(dev).requires_grad_(True)
    w = torch.rand(B, 1, H, H, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))

    def step(module=None, update=update_volume):
        density.grad = None
        rot.grad = None
        xyz.grad = None
        if update:   # an optimiser that USES dL/dvoxel writes the volume every step
            with torch.no_grad():
                density.add_(0.0)
        # DRR.forward from the pose parameters (Euler ZXY, xvr's registration parameterisation): pose -> camera is
        # one HIP launch, then rays -> render, [B,1,H,H]
        img = (module or drr)(rot, xyz, parameterization="euler_angles", convention="ZXY", density=density, **kw)
        if exchange is not None:
            exchange.post_forward(img)
            exchange.pre_backward(density if voxel_grad else None)
        (img * w).sum().backward()
        if exchange is not None:
            exchange.post_backward(density.grad)
            exchange.wait()
        return img

    # algorithmic work of one step, counted by the kernel itself (samples that touch the volume / voxel segments traversed)
    with torch.no_grad():
        work = torch.zeros(1, dtype=torch.int64, device=dev)
        pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
        s, t = drr.detector(pose, None)
        L = (t - s).norm(dim=-1).unsqueeze(1)
        spec = drr.renderer._spec(**kw)
        renderers.render(density.detach(), drr.affine_inverse(s), drr.affine_inverse(t), L, spec, ray_grid_w=H, work=work)
        units = int(work.item())
        del s, t, L
    # Resident data: the renderer keeps a render-ready copy of a STATIC volume next to it (y-pair interleaved for the
    # trilinear march, bricked for Siddon; xvr_amd/renderers.py builds it on the third render of a volume version, 0.7 ms).
    # It belongs to the inputs that are in HBM when the timed region starts, whatever --warmup says.
    with torch.no_grad():
        for _ in range(3):
            drr(rot, xyz, parameterization="euler_angles", convention="ZXY", density=density, **kw)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if exchange is not None:
        dist.barrier()
    renderers.PROFILER = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if exchange is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    events, renderers.PROFILER = renderers.PROFILER, None

    # per-kernel launch durations from the HIP events recorded on the launch stream
    per_kernel = {}
    for name, e0, e1 in events:
        per_kernel.setdefault(name, []).append(e0.elapsed_time(e1))
    kernels = {k: {"launches": len(v), "avg_ms": sum(v) / len(v)} for k, v in per_kernel.items()}
    # the roofline is priced on the dominant RENDER kernel (forward / backward of the chosen renderer); the
    # elementwise helpers (ray generation, from-jacobian) do no taps and are listed under "kernels" only
    render_kernels = [k for k in kernels if k.startswith(renderer)] or list(kernels)
    dominant = max(render_kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"])
    bytes_per_unit = TAP_BYTES_PER_SAMPLE if renderer == "trilinear" else 4
    for k, v in kernels.items():
        tapk = k.startswith(renderer)  # the elementwise from-jacobian kernel does no taps
        v["algorithmic_GBps"] = (units * bytes_per_unit / (v["avg_ms"] * 1e-3) / 1e9) if tapk else None
    dom = kernels[dominant]
    nominal_units = B * H * H * (n_points if renderer == "trilinear" else 0)
    variant = "nx" if renderer == "siddon" and drr_kwargs and (drr_kwargs.get("norm_dims_offset") or drr_kwargs.get("align_corners")) else ""
    for k, v in kernels.items():
        bf = binding_floor(k, units, v["avg_ms"], variant)
        if bf:
            v["binding"] = bf
    roofline = {
        # (`frac` prices SURVEY 8d's algorithmic tap bytes against the HBM peak -- the survey's metric.  The taps are served
        #  on-chip: what the memory side moved is `hbm_physical`, and the unit that really binds the kernel is `binding`.)
        "bound": "hbm", "bound_detail": "algorithmic tap bandwidth (cache-served); binding unit under `binding`", "kernel": dominant,
        "achieved": dom["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": (dom["algorithmic_GBps"] or 0.0) / HBM_PEAK_GBS,
        "frac_of_measured_copy_peak": (dom["algorithmic_GBps"] or 0.0) / HBM_COPY_GBS,
        "traffic": pmc_traffic(dominant, variant),
        "units_per_launch": units, "unit_name": "volume-touching samples" if renderer == "trilinear" else "voxel segments",
        "bytes_per_unit": bytes_per_unit, "avg_launch_ms": dom["avg_ms"],
        "nominal_units_per_launch": nominal_units or None,
    }
    if "binding" in dom:
        roofline["binding"] = dom["binding"]
    # Three ways to price the same launch, side by side: `frac` = counted algorithmic taps (the kernel
    # skips, exactly, the samples that only read padding); `nominal_frac` = SURVEY 8d's nominal B*H*W*N samples -- it can
    # exceed 1 because more than half of them lie outside the volume; `hbm_physical` = what the memory side actually
    # moved (FETCH_SIZE + WRITE_SIZE of the committed rocprofv3 PMC passes; calibrated in round 4,
    # profiles/r04_fetch_calibration.txt: one L2 miss is one 128-byte line tallied at 64 bytes, for 16-byte gathers as for
    # coalesced streams -- `frac_corrected` doubles the fetch side accordingly)
    if nominal_units:
        roofline["nominal_frac"] = nominal_units * bytes_per_unit / (dom["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    if roofline["traffic"]:
        phys = roofline["traffic"] / (dom["avg_ms"] * 1e-3) / 1e9
        roofline["hbm_physical"] = {"GBps": phys, "frac": phys / HBM_PEAK_GBS, "frac_of_measured_copy_peak": phys / HBM_COPY_GBS,
                                    "source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE x 2 as calibrated, WRITE_SIZE; separate passes)"}
    # the forward+backward PAIR priced as one unit (64 B per sample with the voxel gradient,
    # 32 B without), over the summed HIP-event time of every kernel of a step
    kernel_ms = sum(v["avg_ms"] * v["launches"] for v in kernels.values()) / steps
    pair_bytes = units * bytes_per_unit * (2 if voxel_grad else 1)
    roofline["forward_backward_pair"] = {
        "achieved": pair_bytes / (kernel_ms * 1e-3) / 1e9, "unit": "GB/s", "frac": pair_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "kernel_ms_per_step": kernel_ms, "bytes_per_unit": bytes_per_unit * (2 if voxel_grad else 1),
    }
    return {"elapsed": elapsed, "ms_per_step": 1e3 * elapsed / steps, "kernels": kernels, "roofline": roofline, "units": units,
            "step": step, "drr": drr, "spec": spec, "rot": rot, "xyz": xyz, "B": B, "density": density}


def leg_summary(leg, B):
    """What a secondary leg contributes to the JSON line: ms per step, DRRs/s, its kernel table and roofline."""
    return {"ms_per_step": leg["ms_per_step"], "DRRs_per_s": B / (leg["ms_per_step"] * 1e-3), "kernels": leg["kernels"],
            "roofline": leg["roofline"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=116)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--det", type=int, default=256)
    ap.add_argument("--n-points", type=int, default=500)
    ap.add_argument("--renderer", default="trilinear", choices=["trilinear", "siddon"])
    ap.add_argument("--no-voxel-grad", action="store_true", help="pose-only backward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch poses per rank (default); strong: --batch poses in total, split over the ranks")
    ap.add_argument("--cpu-rays", type=int, default=16384, help="rays of one DRR rendered by the CPU baseline")
    ap.add_argument("--no-c1-plumbing", action="store_true", help="skip the whole-DRR configs[0] leg of the CPU baseline")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--update-volume", action="store_true",
                    help="change the volume in place before every step (its version counter moves: no cached render-ready "
                         "layout survives) -- the scenario the voxel gradient exists for")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: initialise the process group (RCCL for --backend nccl) and run the all-gather on one rank")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the secondary legs (volume changing, clip_to_volume, Siddon, pose-only, registration iteration, training step)")
    ap.add_argument("--single-device", action="store_true",
                    help="testing hook: every rank uses cuda:0 (with --backend gloo), to exercise the N > 1 code path on a 1-GPU box")
    ap.add_argument("--check-gather", action="store_true",
                    help="with N > 1 (or --force-dist): rank 0 re-renders EVERY rank's poses itself through the HIP path and compares "
                         "the all-gathered tensor with it value by value (`gather_check` in the JSON line)")
    ap.add_argument("--no-volume-grad-exchange", action="store_true",
                    help="N > 1: leave the all-reduce of the voxel gradient out of the step (the all-gather of the DRRs stays)")
    ap.add_argument("--volume-grad-slabs", type=int, default=4,
                    help="N > 1: the voxel gradient is computed in this many x slabs and each goes into its all-reduce while the next is "
                         "computed (1: the whole gradient first, then bucketed all-reduces)")
    ap.add_argument("--check-volume-grad", action="store_true",
                    help="with N > 1 (or --force-dist): rank 0 renders the UNION of all ranks' poses itself and compares its voxel "
                         "gradient with the all-reduced one (`volume_grad_check` in the JSON line)")
    ap.add_argument("--drr-kwargs", default="", help='JSON of extra DRR / RenderSpec keywords for the headline leg, e.g. \'{"norm_dims_offset": 1}\' '
                                                     "(the recalled knob sets; recorded in config.workload)")
    ap.add_argument("--full-json", default=str(ROOT / "bench_full.json"),
                    help="where the full result goes (kernel tables, floors per candidate unit, spreads): the stdout line carries scalars only")
    ap.add_argument("--c5", action="store_true", help="run the training-step legs at sizes other than 512^3 -> 256^2 (tests)")
    ap.add_argument("--dry-run-collectives", action="store_true",
                    help="allocate the exact tensors of the N-rank step (--gpus N names N; this process is ONE rank), run the rank-local "
                         "part once with the all-gather on a one-rank group of the chosen backend, check the gathered block against "
                         "the render, print a JSON report and exit: the first 8-GPU run then measures instead of debugging")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run_collectives:
        return dry_run_collectives(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the bare command `python bench.py --gpus N`: launch ourselves, one rank per GPU, over the loopback rendezvous
        return self_launch(args.gpus)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus disagree")
    if rank != 0:   # only rank 0 owns stdout (the launcher merges the ranks' streams): whatever a library prints elsewhere goes to stderr
        os.dup2(2, 1)
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        init_group(args.backend, dev, world)

    from xvr_amd.data import make_phantom, read
    from xvr_amd.drr import DRR

    B, H = args.batch, args.det
    if args.scaling == "strong":   # the same B poses whatever the number of ranks: contiguous shares (distributed.shard_bounds)
        from xvr_amd.distributed import shard_bounds
        lo, hi = shard_bounds(B, rank, world)
        B_total, B = B, hi - lo
        if B <= 0:
            raise SystemExit(f"--scaling strong: rank {rank} of {world} has no pose out of {B_total}")
    else:
        lo, B_total = 0, world * B
    vol, _ = make_phantom(args.size, n_ellipsoids=64, seed=0, device=dev)
    subject = read(vol, orientation="AP")
    delx = 1.08821875 * 256 / H  # scripts/v1-submission/pelvis/train/patient_specific.sh:31-33
    if args.scaling == "strong":
        rot, xyz = (t[lo:lo + B] for t in deepfluoro_poses(B_total, seed=0).convert("euler_angles", "ZXY"))
    else:
        rot, xyz = deepfluoro_poses(B, seed=rank).convert("euler_angles", "ZXY")
    Bmax = B if args.scaling == "weak" else -(-B_total // world)
    exchange = Exchange(world, B, Bmax, H, dev, volume_grad=not args.no_volume_grad_exchange and not args.no_voxel_grad,
                        slabs=args.volume_grad_slabs) if use_dist else None

    drr_kwargs = json.loads(args.drr_kwargs) if args.drr_kwargs else None
    leg = render_leg(dev, subject, args.renderer, not args.no_voxel_grad, rot, xyz, H, delx, args.n_points, args.steps, args.warmup,
                     exchange=exchange, update_volume=args.update_volume, drr_kwargs=drr_kwargs)
    elapsed = leg["elapsed"]
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()

    result = {
        "metric": "DRRs/sec (fwd+bwd) 512³ CT→256² detector; achieved HBM GB/s vs peak",
        "value": B_total * args.steps / elapsed, "unit": "DRRs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        # (`dtype` is the arithmetic of the render and of every gradient but one: the voxel gradient's per-voxel SUMS)
        "voxel_gradient_sums": ("none (pose-only backward)" if args.no_voxel_grad else
                                "int32 fixed point per (pose, 16^3 brick), guard-banded (overflow -> NaN + flag); fp32 gather above ~48 samples of a pose per voxel"
                                if args.renderer == "trilinear" else
                                "fp32 (exact index map); int32 fixed point per 16^3 brick, guard-banded, under a non-exact map"),
        "config": {
            "workload": f"single {args.size}^3 CT, {args.renderer} fwd+bwd(pose{'' if args.no_voxel_grad else '+voxel'}), "
                        f"{H}x{H} detector, batch_size={B} per GPU" + (f" ({B_total} in total, strong scaling)" if args.scaling == "strong" else "")
                        + (f", n_points={args.n_points}" if args.renderer == "trilinear" else "")
                        + (f", spec {drr_kwargs}" if drr_kwargs else "")
                        + ("; the volume changes before every step (render-ready copy rebuilt inside the timed region)" if args.update_volume else
                           "; static volume: its render-ready copy (y-pair tiles / bricks) is built before the timed region"),
            "global_batch": B_total, "parallelism": f"pose-sharded x{world}, replicated volume, all-gather of DRRs"
            + ("" if args.no_volume_grad_exchange or args.no_voxel_grad else " + all-reduce of the voxel gradient")
            if world > 1 else "single GPU",
        },
        "roofline": leg["roofline"], "kernels": leg["kernels"],
    }

    if use_dist:
        vg = exchange.volume_grad
        result["volume_grad_exchange"] = {
            "in_the_timed_step": bool(vg), "slabs": exchange.slabs if vg else 0, "buckets": exchange.buckets if vg and exchange.slabs <= 1 else 0,
            "MB_per_rank": (args.size ** 3) * 4 / 1e6 if vg else 0.0,
            "how": ("the backward computes the voxel gradient in x slabs of whole 16^3-brick planes (one launch per slab, same bits as one "
                    "launch); each finished slab goes into an async all_reduce (SUM) while the next is computed -- only the last slab's "
                    "collective is exposed; waited for at the end of the step" if exchange.slabs > 1 else
                    "bucketed async all_reduce (SUM) of density.grad over slabs of the first axis, issued after the backward, waited for "
                    "at the end of the step; not overlapped with the splat"),
        }
        if vg:   # the same step without it, a short loop behind the timed region (same barrier discipline)
            exchange.volume_grad = False
            leg["step"]()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(min(args.steps, 5)):
                leg["step"]()
            torch.cuda.synchronize()
            dist.barrier()
            t_wo = torch.tensor([(time.perf_counter() - t0) / min(args.steps, 5)], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t_wo, op=dist.ReduceOp.MAX)
            exchange.volume_grad = True
            result["volume_grad_exchange"]["ms_per_step_without_it"] = 1e3 * t_wo.item()
            result["volume_grad_exchange"]["value_without_it"] = B_total / t_wo.item()

    if args.check_volume_grad and use_dist and exchange.volume_grad:
        # the all-reduced voxel gradient against the gradient of the UNION batch rendered by rank 0 alone through the same HIP path
        # (the per-pose fixed-point sums are converted and added pose by pose in fp32: the grouping differs, so a tolerance)
        leg["step"]()
        torch.cuda.synchronize()
        summed = leg["density"].grad.detach().clone()
        dist.barrier()
        if rank == 0:
            from xvr_amd.pose import convert  # noqa: F401
            kwr = {"n_points": args.n_points} if args.renderer == "trilinear" else {}
            rr_all, xx_all, ww = [], [], []
            for r in range(world):
                if args.scaling == "strong":
                    from xvr_amd.distributed import shard_bounds
                    l, h = shard_bounds(B_total, r, world)
                    rr, xx = (t[l:h] for t in deepfluoro_poses(B_total, seed=0).convert("euler_angles", "ZXY"))
                else:
                    rr, xx = deepfluoro_poses(args.batch, seed=r).convert("euler_angles", "ZXY")
                rr_all.append(rr); xx_all.append(xx)
                ww.append(torch.rand(rr.shape[0], 1, H, H, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)))
            dens = leg["density"].detach().clone().requires_grad_(True)
            img_u = leg["drr"](torch.cat(rr_all).to(dev), torch.cat(xx_all).to(dev), parameterization="euler_angles", convention="ZXY",
                               density=dens, **kwr)
            (img_u * torch.cat(ww)).sum().backward()
            ref_g = dens.grad
            err = (summed - ref_g).abs().max().item() / max(ref_g.abs().max().item(), 1e-30)
            result["volume_grad_check"] = {"max_rel_err": err, "ok": bool(err <= 2e-6), "ranks": world,
                                           "nonzero_voxels": int((ref_g != 0).sum().item())}
        dist.barrier()

    if args.check_gather and use_dist:
        # the exchange moved the right numbers to the right place: every rank's block of the gathered tensor against a render of
        # that rank's poses done HERE, by this rank alone (the forward is deterministic and independent of the batch's make-up)
        if rank == 0:
            kwr = {"n_points": args.n_points} if args.renderer == "trilinear" else {}
            worst, equal, pad_zero = 0.0, True, True
            for r in range(world):
                if args.scaling == "strong":
                    from xvr_amd.distributed import shard_bounds
                    l, h = shard_bounds(B_total, r, world)
                    rr, xx = (t[l:h] for t in deepfluoro_poses(B_total, seed=0).convert("euler_angles", "ZXY"))
                else:
                    rr, xx = deepfluoro_poses(args.batch, seed=r).convert("euler_angles", "ZXY")
                with torch.no_grad():
                    ref_img = leg["drr"](rr.to(dev), xx.to(dev), parameterization="euler_angles", convention="ZXY", **kwr)
                blk = exchange.gathered[r * Bmax:r * Bmax + rr.shape[0]]
                equal = equal and bool(torch.equal(blk, ref_img))
                worst = max(worst, float((blk - ref_img).abs().max()))
                pad_zero = pad_zero and bool((exchange.gathered[r * Bmax + rr.shape[0]:(r + 1) * Bmax] == 0).all())
            result["gather_check"] = {"equal": equal, "max_abs_diff": worst, "padding_zero": pad_zero, "ranks": world}
        dist.barrier()

    # Secondary legs of the default single-GPU run: every single-GPU configuration of BASELINE.json and every unpinned-knob
    # reading of the headline, measured in the driver's own run.  (i) the headline is conditional on a static volume (cached
    # y-pair copy) and on the unpinned `clip_to_volume` knob (False / per ray / "batch"): three short loops; (ii) C3 = the same
    # step through the Siddon renderer; (iii) the pose-only backward -- the only backward xvr itself requests
    # (registrar/base.py:252, trainer.py:223); (iv) the knob sets SURVEY Appendix A recalls; (v) C4 = one registration iteration
    # at 256^2 and 512^2; (vi) C5 = the render side of one training step, also under the per-ray clip.  The LINE carries their
    # scalars; the kernel tables go to bench_full.json.
    if world == 1 and not args.no_variants and args.renderer == "trilinear" and not args.no_voxel_grad and not use_dist and not drr_kwargs:
        step = leg["step"]

        def timed_loop(n, **kws):
            step(**kws)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                step(**kws)
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t) / n

        variants = {"ms_per_step_volume_changing": timed_loop(5, update=True)}
        for key, knob in (("clip_to_volume_ms_per_step", True), ("clip_batch_ms_per_step", "batch")):
            drr_clip = DRR(subject, 1020.0, H, delx, renderer="trilinear", reverse_x_axis=False, clip_to_volume=knob).to(dev)
            variants[key] = timed_loop(3, module=drr_clip, update=False)
            del drr_clip
        del step
        leg.pop("step")
        nsec = max(3, min(args.steps, 10))
        variants["trilinear_pose_only"] = leg_summary(render_leg(dev, subject, "trilinear", False, rot, xyz, H, delx, args.n_points, nsec, 1), B)
        torch.cuda.empty_cache()
        variants["siddon"] = leg_summary(render_leg(dev, subject, "siddon", True, rot, xyz, H, delx, args.n_points, nsec, 1), B)
        variants["siddon_pose_only"] = leg_summary(render_leg(dev, subject, "siddon", False, rot, xyz, H, delx, args.n_points, nsec, 1), B)
        for v in ("trilinear_pose_only", "siddon", "siddon_pose_only"):
            variants[v].pop("step", None)
        variants["siddon_ms_per_step"] = variants["siddon"]["ms_per_step"]
        variants["pose_only_ms_per_step"] = {"trilinear": variants["trilinear_pose_only"]["ms_per_step"],
                                             "siddon": variants["siddon_pose_only"]["ms_per_step"]}
        torch.cuda.empty_cache()
        # the two knob sets SURVEY.md Appendix A RECALLS for upstream where this build's defaults are the other choice
        # (parity is unpinned: if the pin lands there, these are the numbers) -- trilinear with the per-ray alpha window,
        # Siddon with dims = shape + 1 -- each as the full step and as the pose-only step xvr itself requests
        recalled = {}
        for name, rend, kwd in (("trilinear_clip_per_ray", "trilinear", {"clip_to_volume": True}),
                                ("siddon_dims_plus_1", "siddon", {"norm_dims_offset": 1})):
            full = leg_summary(render_leg(dev, subject, rend, True, rot, xyz, H, delx, args.n_points, nsec, 1, drr_kwargs=kwd), B)
            torch.cuda.empty_cache()
            pose = leg_summary(render_leg(dev, subject, rend, False, rot, xyz, H, delx, args.n_points, nsec, 1, drr_kwargs=kwd), B)
            torch.cuda.empty_cache()
            recalled[name] = {"spec": kwd, "ms_per_step": full["ms_per_step"], "DRRs_per_s": full["DRRs_per_s"],
                              "pose_only_ms_per_step": pose["ms_per_step"], "pose_only_DRRs_per_s": pose["DRRs_per_s"],
                              "kernels": full["kernels"], "roofline": full["roofline"]}
        variants["recalled_knobs"] = recalled
        # (2048^2 X-ray at 0.136 mm, scales "8,4": 256^2 then 512^2, registrar/base.py:402-407; scaled with --det for the tiny test runs)
        c4_delx = 0.1360 * 8 * 256 / H
        # (every parameterisation the reference's registrar offers runs the device-resident loop, and Equalize runs inside it)
        c4_extra = tuple(dict(parameterization=p) for p in ("se3_log_map", "axis_angle", "quaternion", "quaternion_adjugate", "rotation_6d", "rotation_10d")) + \
                   (dict(equalize=True),)
        variants["c4_register_ms_per_pose_iteration"] = benchlib.c4_register(dev, subject, sizes=((H, c4_delx), (2 * H, c4_delx / 2)), extra=c4_extra)
        torch.cuda.empty_cache()
        if args.size == 512 and H == 256 or args.c5:
            c5 = benchlib.c5_train_step(dev, size=args.size, B=B, H=H)
            variants["c5_train_step_ms"] = c5["ms_per_step"]
            variants["c5_train_step"] = c5
            torch.cuda.empty_cache()
            # the same training step if upstream's alpha rule is the per-ray window (mask = seg goes through it too, trainer.py:288)
            c5c = benchlib.c5_train_step(dev, size=args.size, B=B, H=H, n=5, warm=2, drr_kwargs={"clip_to_volume": True})
            variants["c5_train_step_clip_ms"] = c5c["ms_per_step"]
            variants["c5_train_step_clip"] = c5c
        result["variants"] = variants
        result["ms_per_step_volume_changing"] = variants["ms_per_step_volume_changing"]
        result["clip_to_volume_ms_per_step"] = variants["clip_to_volume_ms_per_step"]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(vol, leg["drr"], leg["rot"], leg["xyz"], leg["spec"], args)
    if use_dist:   # (before the line: whatever the backend says while it shuts down must not follow it on stdout)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(result, args.full_json)


def _r(x, digits=5):
    """A float of the short line: `digits` significant digits (the timed headline keeps its full precision)."""
    return float(f"{x:.{digits}g}") if isinstance(x, float) else x


def compact_line(full):
    """The ONE line the driver parses, from the full result: the contract's keys, `roofline` and `cpu_baseline` as objects of
    scalars, the headline under the other readings of the unpinned knobs as top-level scalars, per-call averages and the secondary
    legs as flat dictionaries of scalars.  Kernel tables, floors per candidate unit, sources and spreads stay in the full result
    (bench_full.json and stderr).  Round 5's line was 23 KB and the driver could not parse it: this one is held below 8 KB by
    tests/test_bench_contract.py."""
    B = full["config"]["global_batch"] / max(full["n_gpus"], 1)
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data", "voxel_gradient_sums", "config")}
    v = full.get("variants")
    if v:   # the same step under the other readings of what is not pinned, in the headline's unit
        out["value_clip_per_ray"] = _r(B / (v["clip_to_volume_ms_per_step"] * 1e-3))
        out["value_clip_batch"] = _r(B / (v["clip_batch_ms_per_step"] * 1e-3))
        out["value_volume_changing"] = _r(B / (v["ms_per_step_volume_changing"] * 1e-3))
    r = full["roofline"]
    ro = {k: _r(r[k], 6) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "units_per_launch", "unit_name",
                                   "bytes_per_unit", "avg_launch_ms") if k in r}
    if "nominal_frac" in r:
        ro["nominal_frac"] = _r(r["nominal_frac"])
    if r.get("binding"):
        ro["binding"] = {k: _r(r["binding"][k]) for k in ("unit", "floor_ms", "frac")}
    if r.get("hbm_physical"):
        ro["hbm_physical"] = {k: _r(r["hbm_physical"][k]) for k in ("GBps", "frac")}
    p = r["forward_backward_pair"]
    ro["step_pair"] = {"bytes_per_unit": p["bytes_per_unit"], "achieved": _r(p["achieved"]), "frac": _r(p["frac"]), "kernel_ms_per_step": _r(p["kernel_ms_per_step"])}
    out["roofline"] = ro
    out["kernels_ms"] = {k: _r(x["avg_ms"], 4) for k, x in full["kernels"].items()}
    for k in ("volume_grad_exchange", "gather_check", "volume_grad_check"):
        if k in full:
            out[k] = {a: _r(b) for a, b in full[k].items() if not isinstance(b, str)}
    c = full.get("cpu_baseline")
    if c:
        out["cpu_baseline"] = {"value": _r(c["value"]), "unit": c["unit"], "cores": c["cores"], "threads": c["threads"], "kind": c["kind"],
                               "sample": c["sample"]}
        if "c1_plumbing" in c:
            out["cpu_baseline"]["c1_plumbing_128x128_DRRs_per_s"] = _r(c["c1_plumbing"]["value"])
    if v:
        rk, c4 = v["recalled_knobs"], v["c4_register_ms_per_pose_iteration"]
        flat = {
            "volume_changing_ms": v["ms_per_step_volume_changing"], "clip_per_ray_ms": v["clip_to_volume_ms_per_step"],
            "clip_batch_ms": v["clip_batch_ms_per_step"], "siddon_ms": v["siddon_ms_per_step"],
            "siddon_nx_ms": rk["siddon_dims_plus_1"]["ms_per_step"],
            "pose_only_ms": {"trilinear": v["pose_only_ms_per_step"]["trilinear"], "siddon": v["pose_only_ms_per_step"]["siddon"],
                             "trilinear_clip_per_ray": rk["trilinear_clip_per_ray"]["pose_only_ms_per_step"],
                             "siddon_nx": rk["siddon_dims_plus_1"]["pose_only_ms_per_step"]},
            "siddon_kernels_ms": {k: x["avg_ms"] for k, x in v["siddon"]["kernels"].items() if k.startswith("siddon")},
            "siddon_nx_kernels_ms": {k: x["avg_ms"] for k, x in rk["siddon_dims_plus_1"]["kernels"].items() if k.startswith("siddon")},
            "clip_per_ray_kernels_ms": {k: x["avg_ms"] for k, x in rk["trilinear_clip_per_ray"]["kernels"].items() if k.startswith("trilinear")},
            "c4_ms_per_iter": {k: c4[k]["single"] for k in c4},
            "c4_ms_per_pose_iter_batched8": {k: c4[k]["batched8"] for k in c4},
        }
        if "c5_train_step_ms" in v:
            flat["c5_step_ms"] = v["c5_train_step_ms"]
            flat["c5_step_ms_clip"] = v["c5_train_step_clip_ms"]

        def rnd(d):
            return {k: (rnd(x) if isinstance(x, dict) else _r(x, 4)) for k, x in d.items()}
        out["variants"] = rnd(flat)
    out["full"] = "bench_full.json next to bench.py, and on stderr"
    return out


def _no_constants(name):
    raise ValueError(f"{name} in the JSON line")


def emit(full, full_json):
    """Full result -> `full_json` (and stderr, which the driver pulls as n1.err); the compact line -> stdout, LAST."""
    text = json.dumps(full)
    try:
        Path(full_json).write_text(text + "\n")
    except OSError as e:
        print(f"bench.py: could not write {full_json}: {e}", file=sys.stderr)
    print("bench_full " + text, file=sys.stderr, flush=True)
    line = json.dumps(compact_line(full), allow_nan=False, separators=(",", ":"))
    json.loads(line, parse_constant=_no_constants)
    if len(line) >= 8192:
        raise SystemExit(f"bench.py: the JSON line is {len(line)} bytes; the driver needs it below 8 KB")
    sys.stderr.flush()
    flush_c_stdio()
    # (a fresh line whatever a library left unterminated on stdout: the driver reads the LAST line)
    sys.stdout.write("\n" + line + "\n")
    sys.stdout.flush()


def flush_c_stdio():
    """RCCL writes its version banner through C stdio, which is fully buffered when stdout is a pipe and comes out when the PROCESS
    exits -- behind the JSON line Python flushed long before (seen in round 6: `Librccl path : ...` was the last stdout line of the
    one-rank RCCL run).  Every rank empties the C buffers right after the group's first collective, and rank 0 again before the line."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def self_launch(n):
    """`python bench.py --gpus N` outside a launcher: re-exec the same command line under torch.distributed.run (one process per
    GPU, rendezvous on 127.0.0.1 at a free port), pass its output through and exit with its status."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def init_group(backend, dev, world):
    kw_pg = {}
    if world == 1 and "RANK" not in os.environ:   # --force-dist outside a launcher: a one-rank group on the loopback
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        kw_pg = dict(init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev, **kw_pg)
    else:
        dist.init_process_group(backend, **kw_pg)
    # the first collective creates the communicator (and makes RCCL say what it has to say): out with it now, on every rank
    t = torch.zeros(1, device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    flush_c_stdio()


def dry_run_collectives(args):
    """`--gpus N --dry-run-collectives` on ONE GPU: this process plays rank 0 of N.  It allocates the tensors of the N-rank step
    exactly as a rank would (its share of the poses, the [N * Bmax, 1, H, H] gather buffer, the padded send buffer of a ragged
    strong split), initialises a one-rank group of --backend (nccl = RCCL), runs the step -- the async all_gather_into_tensor
    between forward and backward included -- and checks that the block the gather wrote equals this rank's render and that
    nothing else of the buffer was touched.  What it cannot check is more than one rank over xGMI."""
    N = max(args.gpus, 1)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    init_group(args.backend, dev, 1)
    from xvr_amd.data import make_phantom, read
    from xvr_amd.distributed import shard_bounds

    B, H = args.batch, args.det
    report = {"dry_run_collectives": True, "as_world_size": N, "backend": args.backend, "scaling": args.scaling, "legs": []}
    vol, _ = make_phantom(args.size, n_ellipsoids=64, seed=0, device=dev)
    subject = read(vol, orientation="AP")
    delx = 1.08821875 * 256 / H
    for rank in sorted({0, N - 1}):      # the first rank and the last (the ragged share of a strong split)
        if args.scaling == "strong":
            lo, hi = shard_bounds(B, rank, N)
            Bl, B_total = hi - lo, B
            rot, xyz = (t[lo:hi] for t in deepfluoro_poses(B_total, seed=0).convert("euler_angles", "ZXY"))
        else:
            Bl, B_total = B, N * B
            rot, xyz = deepfluoro_poses(B, seed=rank).convert("euler_angles", "ZXY")
        Bmax = Bl if args.scaling == "weak" else -(-B_total // N)
        # a one-rank group gathers into a buffer of ONE share; the N-rank buffer is allocated as well (same bytes a rank holds)
        full = torch.full((N * Bmax, 1, H, H), float("nan"), device=dev)
        ex = Exchange(1, Bl, Bmax, H, dev)
        leg = render_leg(dev, subject, args.renderer, not args.no_voxel_grad, rot, xyz, H, delx, args.n_points, 1, 0, exchange=ex)
        with torch.no_grad():
            img = leg["drr"](leg["rot"], leg["xyz"], parameterization="euler_angles", convention="ZXY",
                             **({"n_points": args.n_points} if args.renderer == "trilinear" else {}))
        # the voxel gradient's bucketed all-reduce ran inside that step (on the one-rank group: a sum over one rank); the same step
        # without it must leave the very same gradient
        g_with = leg["density"].grad.detach().clone() if not args.no_voxel_grad else None
        ex.volume_grad = False
        leg["step"]()
        torch.cuda.synchronize()
        ok_grad = bool(g_with is None or torch.equal(g_with, leg["density"].grad))
        full[rank * Bmax:rank * Bmax + Bmax].copy_(ex.gathered)
        ok_block = bool(torch.equal(ex.gathered[:Bl], img))
        ok_pad = bool((ex.gathered[Bl:] == 0).all()) if Bmax > Bl else True
        ok_rest = bool(torch.isnan(full[:rank * Bmax]).all() and torch.isnan(full[(rank + 1) * Bmax:]).all())
        report["legs"].append({"as_rank": rank, "poses": Bl, "share_padded_to": Bmax, "gather_buffer_MB": full.numel() * 4 / 1e6,
                               "send_MB": Bmax * H * H * 4 / 1e6, "gathered_block_equals_render": ok_block, "padding_is_zero": ok_pad,
                               "rest_untouched": ok_rest, "volume_grad_allreduce_identity": ok_grad,
                               "volume_grad_slabs": 0 if args.no_voxel_grad else ex.slabs, "volume_grad_MB": 0.0 if args.no_voxel_grad else args.size ** 3 * 4 / 1e6,
                               "ms_step": leg["ms_per_step"]})
        del full, ex, leg
        torch.cuda.empty_cache()
    report["ok"] = all(l["gathered_block_equals_render"] and l["padding_is_zero"] and l["rest_untouched"] and l["volume_grad_allreduce_identity"]
                       for l in report["legs"])
    print(json.dumps(report))
    dist.destroy_process_group()
    if not report["ok"]:
        raise SystemExit(1)


def host_cpu():
    """(model name, physical cores, hardware threads) of this host, from /proc/cpuinfo."""
    model, cores, threads = "unknown", set(), 0
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            key, _, val = line.partition(":")
            key, val = key.strip(), val.strip()
            if key == "model name":
                model = val
            elif key == "processor":
                threads += 1
            elif key == "physical id":
                phys = val
            elif key == "core id":
                core = val
            elif not key and phys is not None:
                cores.add((phys, core))
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or threads or (os.cpu_count() or 1)), (threads or (os.cpu_count() or 1))


def _timed_reps(run, min_reps=3, budget_s=12.0, max_reps=8):
    run()  # warm-up (page in the volume, thread pool)
    times = []
    t_start = time.perf_counter()
    while len(times) < min_reps or (time.perf_counter() - t_start < budget_s and len(times) < max_reps):
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    return times


def cpu_baseline(vol, drr, rot, xyz, spec, args):
    """The oracle (torch-ops restatement of the reference's CPU render path: linspace -> grid_sample ->
    sum, autograd backward to pose and voxels) on this host's cores.  Two bounded legs:
      * `value`: the benchmark's own workload -- `cpu_rays` centre rays of pose 0, scaled to whole DRRs (same unit as the
        headline; >= 3 repetitions, the median is reported);
      * `c1_plumbing`: BASELINE.json configs[0] run WHOLE, as the reference would on its CPU path: DeepFluoro geometry,
        128 x 128 detector, delx 2.1764375, batch_size 4, n_points 500 (scripts/deepfluoro/train/de_novo.sh:24-32), forward +
        backward to pose and voxels, in 128 x 128 DRRs per second."""
    import dataclasses
    import statistics

    from oracle.diffdrr_restated import RenderSpec as OSpec, drr_from_pose, render as oracle_render
    from xvr_amd.pose import convert

    model, ncores, nthreads = host_cpu()
    torch.set_num_threads(ncores)     # one thread per physical core (SMT siblings only add contention to grid_sample's backward)
    H = args.det
    nrays = min(args.cpu_rays, H * H)
    with torch.no_grad():
        pose = convert(rot[:1].detach(), xyz[:1].detach(), parameterization="euler_angles", convention="ZXY")
        s, t = drr.detector(pose, None)
        L = (t - s).norm(dim=-1).unsqueeze(1)
        s, t = drr.affine_inverse(s), drr.affine_inverse(t)
    # a centred block of detector rows (the border rows would miss the volume and flatter the CPU)
    r0 = (H * H - nrays) // 2
    vol_c = vol.detach().cpu().requires_grad_(not args.no_voxel_grad)
    s_c = s.cpu().requires_grad_(True)
    t_c = t[:, r0:r0 + nrays].cpu().requires_grad_(True)
    L_c = L[..., r0:r0 + nrays].cpu()
    ospec = OSpec(**dataclasses.asdict(spec))

    def run():
        vol_c.grad = None
        out = oracle_render(vol_c, s_c, t_c, L_c, ospec, chunk=16384)
        out.sum().backward()

    times = _timed_reps(run)
    dt = statistics.median(times)
    result = {
        "value": (nrays / (H * H)) / dt, "unit": "DRRs/s", "cores": ncores, "threads": torch.get_num_threads(), "kind": "port",
        "sample": f"{nrays} centre rays of one {H}x{H} DRR of the same workload, fwd+bwd, median of {len(times)} reps "
                  f"({min(times):.2f}-{max(times):.2f} s; torch {torch.__version__} CPU ops, {torch.get_num_threads()} threads on "
                  f"{ncores} physical cores, {model})",
    }
    if args.renderer == "trilinear" and not args.no_c1_plumbing:
        # configs[0], whole DRRs, over a volume of the benchmark's size class cropped to what fits a CPU run in seconds
        Hc, Bc = 128, 4
        sub_vol = vol.detach().cpu()[::2, ::2, ::2].contiguous()   # 2 mm voxels: the same field of view, 1/8 of the voxels
        affine = torch.diag(torch.tensor([2.0, 2.0, 2.0, 1.0]))
        affine[:3, 3] = -(affine[:3, :3] @ ((torch.tensor(sub_vol.shape, dtype=torch.float32) - 1) / 2))
        rc, tc = deepfluoro_poses(Bc, seed=0).convert("euler_angles", "ZXY")
        rc.requires_grad_(True)
        tc.requires_grad_(True)
        v1 = sub_vol.clone().requires_grad_(True)
        c1spec = dataclasses.replace(ospec, n_points=500)

        def run_c1():
            v1.grad = rc.grad = tc.grad = None
            pose_c = convert(rc, tc, parameterization="euler_angles", convention="ZXY")
            img = drr_from_pose(v1, affine, pose_c.matrix, Hc, Hc, 1020.0, 2.1764375, 2.1764375, 0.0, 0.0, c1spec,
                                orientation="AP", reverse_x_axis=True, chunk=8192)
            img.sum().backward()

        t1 = _timed_reps(run_c1, budget_s=8.0)
        d1 = statistics.median(t1)
        result["c1_plumbing"] = {
            "value": Bc / d1, "unit": "128x128 DRRs/s", "reps": len(t1), "seconds_per_batch": d1,
            "config": f"DeepFluoro geometry, trilinear, 128x128, delx 2.1764375, sdd 1020, batch_size {Bc}, n_points 500, "
                      f"{tuple(sub_vol.shape)} volume at 2 mm, fwd+bwd(pose+voxel), whole DRRs",
        }
    return result


if __name__ == "__main__":
    main()

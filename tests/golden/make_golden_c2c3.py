"""The WHOLE benchmark batch through the oracle, once: BASELINE.json configs[1] / configs[2] -- the 512^3
phantom and the 116 DeepFluoro poses of bench.py, 256 x 256 detector, trilinear (n_points = 500) and Siddon -- rendered by
oracle/diffdrr_restated.py in chunks on the CPU, every pixel of every pose, with the gradient of a weighted image sum w.r.t. the six
pose parameters.  What is committed (tests/golden/c2c3_oracle_batch.npz, ~2 MB; the full images would be 60 MB):

  * 4096 pixels of every pose at fixed seeded positions, float32                       (pixel-exact comparison)
  * the sums over all 256 tiles of 16 x 16 pixels of every pose, float64               (every pixel of every pose is in one)
  * the image's maximum and sum per pose
  * d (sum_pixels w_b * img_b) / d (rot_b, xyz_b), Euler ZXY, per pose, float64        (the pose gradient over all pixels)

tests/test_configs.py::test_benchmark_batch_against_the_oracle_fixture compares the HIP path with it on the GPU.

    python tests/golden/make_golden_c2c3.py [--threads 6] [--poses 0:116] [--renderers trilinear,siddon] [--out DIR]

Resumable: every pose is written to DIR (default tests/golden/_c2c3_parts, git-ignored) as it finishes; the last step packs the
parts into the .npz.  ~25 s (trilinear) + ~45 s (Siddon) per pose on 8 cores.
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

SIZE, H, B, N_POINTS, SDD, DELX = 512, 256, 116, 500, 1020.0, 1.08821875
N_SAMPLED = 4096


def weights(b):
    return torch.from_numpy(np.random.default_rng(1000 + b).uniform(0.0, 1.0, size=(1, 1, H, H))).to(torch.float32)


def sampled_pixels():
    return np.sort(np.random.default_rng(77).choice(H * H, size=N_SAMPLED, replace=False))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--poses", default=f"0:{B}")
    ap.add_argument("--renderers", default="trilinear,siddon")
    ap.add_argument("--out", default=str(Path(__file__).resolve().parent / "_c2c3_parts"))
    ap.add_argument("--pack-only", action="store_true")
    ap.add_argument("--variant", default="", choices=["", "clip", "nx"],
                    help="'clip': the trilinear render with the per-ray alpha window SURVEY.md Appendix A recalls for upstream "
                         "(clip_to_volume = True; bench.py's recalled_knobs) -> tests/golden/c2c3_oracle_batch_clip.npz, keys trilinear_clip_*.  "
                         "'nx': Siddon under the recalled index map dims = shape + 1 on a 511^3 phantom (an even-sized axis carries the map's "
                         "structural tie, tests/conftest.py::has_structural_tie) -> c2c3_oracle_batch_nx.npz, keys siddon_nx_*; run it with "
                         "--dtype float64 AND --dtype float32: the float32 run's pixels and gradients are the test's yardstick")
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"],
                    help="arithmetic of the oracle run.  Siddon's pose gradient is a sum of jumps of a piecewise-constant integrand over "
                         "65536 rays: on the phantom's sharp ellipsoid surfaces the float32 oracle is 1e-2 from its own float64 run, so the "
                         "committed Siddon fixture is the float64 run (tests/test_fuzz_large.py makes the same choice); parts of both "
                         "kinds may coexist, the packer prefers float64 ones")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    out = Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    lo, hi = (int(x) for x in args.poses.split(":"))
    renderers = args.renderers.split(",") if not args.variant else ["trilinear" if args.variant == "clip" else "siddon"]
    size = 511 if args.variant == "nx" else SIZE
    tag = lambda renderer: renderer + ("_" + args.variant if args.variant else "")   # noqa: E731
    if not args.pack_only:
        from oracle.diffdrr_restated import RenderSpec, drr_from_pose
        from xvr_amd.data import make_phantom, read
        from xvr_amd.pose import convert
        from xvr_amd.training import get_random_pose

        dt = getattr(torch, args.dtype)
        vol, _ = make_phantom(size, n_ellipsoids=64, seed=0)
        affine = read(vol, orientation="AP").affine.to(dt)
        vol = vol.to(dt)
        g = torch.Generator().manual_seed(0)   # (bench.py::deepfluoro_poses(116, seed=0))
        rot, xyz = get_random_pose(135.0, 225.0, -45.0, 45.0, -15.0, 15.0, -150.0, 150.0, 450.0, 1000.0, -150.0, 150.0, B,
                                   generator=g).convert("euler_angles", "ZXY")
        pix = torch.from_numpy(sampled_pixels())
        for renderer in renderers:
            spec = RenderSpec(renderer=renderer, n_points=N_POINTS, clip_to_volume=args.variant == "clip",
                              norm_dims_offset=1 if args.variant == "nx" else 0)
            for b in range(lo, hi):
                part = out / (f"{tag(renderer)}_{b:03d}.npz" if args.dtype == "float32" else f"{tag(renderer)}_{b:03d}_f64.npz")
                if part.exists():
                    continue
                t0 = time.time()
                r = rot[b:b + 1].clone().to(dt).requires_grad_(True)
                x = xyz[b:b + 1].clone().to(dt).requires_grad_(True)
                pose = convert(r, x, parameterization="euler_angles", convention="ZXY")
                img = drr_from_pose(vol, affine, pose.matrix, H, H, SDD, DELX, DELX, 0.0, 0.0, spec, orientation="AP",
                                    reverse_x_axis=False, chunk=8192 if renderer == "trilinear" else 2048)
                (img * weights(b).to(dt)).sum().backward()
                im = img.detach()[0, 0]
                tiles = im.double().reshape(16, 16, 16, 16).sum(dim=(1, 3))
                np.savez(part, pixels=im.reshape(-1)[pix].float().numpy(), tiles=tiles.numpy(), imax=im.max().item(), isum=im.double().sum().item(),
                         grad=torch.cat([r.grad, x.grad], dim=-1).double().numpy()[0], rot=rot[b].numpy(), xyz=xyz[b].numpy())
                print(f"{tag(renderer)} pose {b}: {time.time() - t0:.1f} s, max {im.max().item():.3f}", flush=True)
    packed = {"pixel_index": sampled_pixels()}
    for renderer in (("trilinear", "siddon") if not args.variant else (tag(renderers[0]),)):
        parts = [out / f"{renderer}_{b:03d}_f64.npz" for b in range(B)]
        if not all(p.exists() for p in parts):
            parts = [out / f"{renderer}_{b:03d}.npz" for b in range(B)]
        packed[f"{renderer}_oracle_dtype"] = np.array("float64" if parts[0].name.endswith("_f64.npz") else "float32")
        if not all(p.exists() for p in parts):
            print(f"{renderer}: {sum(p.exists() for p in parts)} of {B} poses done; not packed")
            continue
        loaded = [np.load(p) for p in parts]
        for key in ("pixels", "tiles", "imax", "isum", "grad", "rot", "xyz"):
            packed[f"{renderer}_{key}"] = np.stack([d[key] for d in loaded])
        f32 = [out / f"{renderer}_{b:03d}.npz" for b in range(B)]
        if parts[0].name.endswith("_f64.npz") and all(p.exists() for p in f32):
            # the float32 run's gradient next to the float64 one: where the two disagree the gradient is ill-conditioned in float32
            # (Siddon, pose 77: 9e-2 of the largest entry) and the test allows the HIP path as much
            packed[f"{renderer}_grad_f32"] = np.stack([np.load(p)["grad"] for p in f32])
            if args.variant == "nx":   # (and its pixels: under a non-exact map two float32 evaluations differ by whole segments on some rays)
                packed[f"{renderer}_pixels_f32"] = np.stack([np.load(p)["pixels"] for p in f32])
    if len(packed) > 1:
        np.savez_compressed(Path(__file__).resolve().parent / ("c2c3_oracle_batch.npz" if not args.variant else f"c2c3_oracle_batch_{args.variant}.npz"), **packed)
        print({k: v.shape for k, v in packed.items()})


if __name__ == "__main__":
    main()

"""Generate tests/golden/*.npz: seeded tiny inputs + the expected outputs of the oracle.

The reference's own implementation of this path (diffdrr==0.6.0) cannot be imported in this
container, and the reference holds no golden vectors, so these fixtures are
produced by the build's own restatement (oracle/diffdrr_restated.py, float32 torch ops) and
cross-checked against the independent float64 scalar oracle (oracle/drr_scalar.c) at generation
time.  They pin the oracle against regressions and travel to the GPU box as data.

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from conftest import clip_mask_tie_free, face_nudges, make_case  # noqa: E402
from oracle import scalar  # noqa: E402
from oracle.diffdrr_restated import RenderSpec, render  # noqa: E402

OUT = Path(__file__).resolve().parent

CASES = {
    "trilinear_default": dict(renderer="trilinear", n_points=60),
    "trilinear_corner_nminus1": dict(renderer="trilinear", n_points=45, voxel_shift=0.0, step_mode="n_minus_1"),
    # (38 samples: 37 intervals share no factor with 20 / 24 / 28 voxels, so no INTERIOR sample of a ray that spans an axis sits on
    #  a label boundary -- conftest.clip_mask_tie_free; round 5's fixture had 40 and its mask half was never compared)
    "trilinear_clip": dict(renderer="trilinear", n_points=38, clip_to_volume=True),
    # the same window drawn in by 3 % at either end: no sample on a face, so the mask half is well-posed as it stands
    "trilinear_clip_inset": dict(renderer="trilinear", n_points=38, clip_to_volume=True, near=0.03, far=0.97),
    "trilinear_recalled_dims": dict(renderer="trilinear", n_points=50, norm_dims_offset=-1),
    "siddon_default": dict(renderer="siddon"),
    "siddon_corner": dict(renderer="siddon", voxel_shift=0.0),
}


def main():
    torch.set_num_threads(4)
    case = make_case()
    for name, kw in CASES.items():
        spec = RenderSpec(**kw)
        arrays = {k: case[k].numpy() for k in ("volume", "mask", "source", "target", "img")}
        for tag, mask in (("nomask", None), ("mask", case["mask"])):
            vol = case["volume"].clone().requires_grad_(True)
            src = case["source"].clone().requires_grad_(True)
            tgt = case["target"].clone().requires_grad_(True)
            img = case["img"].clone().requires_grad_(True)
            out = render(vol, src, tgt, img, spec, mask)
            g = torch.Generator().manual_seed(1)
            w = torch.rand(out.shape, generator=g)
            (out * w).sum().backward()
            ref64 = scalar.render(case["volume"], case["source"], case["target"], case["img"], spec, mask)
            err = np.abs(out.detach().double().numpy() - ref64).max() / max(np.abs(ref64).max(), 1e-12)
            if name == "trilinear_clip" and tag == "mask":
                # clip + mask puts the first / last sample of every ray exactly on a face of the volume, where the nearest-label
                # lookup is a rounding tie (the voxel inside, or the zero padding outside = channel 0).  The fixture holds all four
                # readings (first: in / out) x (last: in / out), named explicitly by a nudge of 1e-3 voxel along the face's normal; a
                # consumer matches every ray to one of them.  The float64 scalar oracle must do so too:
                assert clip_mask_tie_free(case["volume"].shape, spec.n_points)
                p0, p1 = face_nudges(case["source"], case["target"], case["volume"].shape, spec)
                faces = []
                with torch.no_grad():
                    for s0 in (1.0, -1.0):
                        for s1 in (1.0, -1.0):
                            faces.append(render(case["volume"], case["source"], case["target"], case["img"], spec, mask, label_nudge=(s0 * p0, s1 * p1)))
                faces = torch.stack(faces)                                           # [4, B, C, n]
                dev = (torch.from_numpy(ref64).float()[None] - faces).abs().amax(dim=2).amin(dim=0)      # best reading per ray
                err = dev.max().item() / max(np.abs(ref64).max(), 1e-12)
                assert err < 2e-5, (name, tag, err)
                arrays["out_mask_faces"] = faces.numpy()
            else:
                assert err < 2e-5, (name, tag, err)
            arrays.update({
                f"out_{tag}": out.detach().numpy(), f"w_{tag}": w.numpy(),
                f"gvol_{tag}": vol.grad.numpy(), f"gsrc_{tag}": src.grad.numpy(),
                f"gtgt_{tag}": tgt.grad.numpy(), f"gimg_{tag}": img.grad.numpy(),
            })
            print(f"{name:28s} {tag:7s} max|out|={np.abs(ref64).max():9.4f} rel.err vs f64 scalar={err:.2e}")
        arrays["spec_keys"] = np.array(list(kw.keys()))
        arrays["spec_vals"] = np.array([str(v) for v in kw.values()])
        np.savez_compressed(OUT / f"{name}.npz", **arrays)


if __name__ == "__main__":
    main()

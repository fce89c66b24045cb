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

from conftest import make_case  # noqa: E402
from oracle import scalar  # noqa: E402
from oracle.diffdrr_restated import RenderSpec, render  # noqa: E402

OUT = Path(__file__).resolve().parent

CASES = {
    "trilinear_default": dict(renderer="trilinear", n_points=60),
    "trilinear_corner_nminus1": dict(renderer="trilinear", n_points=45, voxel_shift=0.0, step_mode="n_minus_1"),
    "trilinear_clip": dict(renderer="trilinear", n_points=40, clip_to_volume=True),
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
            if not (name == "trilinear_clip" and tag == "mask"):
                # (clip + mask puts the first/last sample exactly on the volume face, where the
                #  nearest-label lookup is decided by rounding noise)
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

"""Golden vectors for the X-ray pixel pipeline, produced by the REFERENCE's own function in this container.

``/root/reference/src/xvr/io/xray.py`` cannot be imported here (its module imports pydicom and torchvision, neither
installed), but the function that does the arithmetic, ``_preprocess_xray`` (xray.py:93-129), needs only torch and one helper
from torchvision, ``center_crop``.  This script compiles THAT function definition out of the reference file (nothing of its
text is stored anywhere) and runs it with the one missing name bound to the documented torchvision behaviour -- a centred
``[..., top:top + h, left:left + w]`` window with ``top = int(round((H - h) / 2.0))``.  That binding is the single stand-in,
it matters only for odd crops, and the vectors record which cases those are (``odd_crop``).

    python tests/golden/make_golden_xray.py       (needs /root/reference; the .npz it writes is committed)
"""
import ast
from pathlib import Path
from typing import Callable

import numpy as np
import torch

REF = Path("/root/reference/src/xvr/io/xray.py")


def _torchvision_center_crop(img, size):
    th, tw = size
    h, w = img.shape[-2:]
    top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
    return img[..., top:top + th, left:left + tw]


def reference_function():
    tree = ast.parse(REF.read_text())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "_preprocess_xray")
    ns = {"torch": torch, "Callable": Callable, "center_crop": _torchvision_center_crop}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), str(REF), "exec"), ns)
    return ns["_preprocess_xray"]


def main():
    ref = reference_function()
    g = torch.Generator().manual_seed(11)
    out, k = {}, 0
    for (h, w), crop in (((40, 36), 0), ((40, 36), 6), ((41, 37), 5), ((40, 36), 5), ((33, 48), 10)):
        for background in (False, True):
            for linearize in (False, True):
                # 12-bit-like integer counts: the mode is well defined (a flat background level covers a third of the image)
                img = torch.randint(200, 4000, (1, 1, h, w), generator=g).float()
                img[0, 0, : h // 3] = 3900.0
                res = ref(img.clone(), crop, background, linearize, "max")
                out[f"c{k}_in"], out[f"c{k}_out"] = img.numpy(), res.numpy()
                out[f"c{k}_cfg"] = np.array([crop, int(background), int(linearize), int(((h - (h - crop)) % 2) or ((w - (w - crop)) % 2))])
                k += 1
    out["n_single"] = np.array(k)
    frames = torch.randint(100, 3000, (1, 1, 5, 24, 28), generator=g).float()
    for j, how in enumerate(("max", "sum", 3, None)):
        res = ref(frames.clone(), 4, False, True, how)
        out[f"m{j}_out"] = res.numpy()
    out["m_in"] = frames.numpy()
    np.savez_compressed(Path(__file__).resolve().parent / "xvr_reference_xray.npz", **out)
    print({key: v.shape for key, v in out.items() if key.endswith("_out")})


if __name__ == "__main__":
    main()

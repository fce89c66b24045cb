"""Per-magnitude accuracy of the fixed-point voxel gradient (k_trilinear_splat_b16) against the float64 oracle, next to the
fp32 table gather: the table of HISTORY.md section 4.1 / include/xvr_drr.h.  Run on the GPU box.

    python tools/splat_accuracy.py [--full]      (--full adds the benchmark size: 512^3 -> 256^2, two poses, ~1 min of CPU)
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import accuracy_by_magnitude, format_accuracy_table  # noqa: E402
from test_splat import fixed_point_accuracy_case  # noqa: E402

for which in ("ordinary", "fine-detector"):
    ref, got = fixed_point_accuracy_case(which)
    print(f"### {which}\n" + format_accuracy_table(accuracy_by_magnitude(ref, got), ["splat", "gather"]) + "\n", flush=True)
if "--full" in sys.argv:
    from test_configs import full_size_accuracy_case  # noqa: E402
    ref, got = full_size_accuracy_case()
    print("### 512^3 -> 256^2, two benchmark poses\n" + format_accuracy_table(accuracy_by_magnitude(ref, got), ["splat", "gather"]), flush=True)

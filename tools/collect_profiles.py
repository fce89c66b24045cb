"""Copy what tools/profile_round.sh left under gpurun_out/ (scratch, merged back from the GPU box) into profiles/ (tracked):
    python tools/collect_profiles.py r05
-> profiles/<round>_{trilinear,pose_only,siddon,siddon_nx}_{rocprof_summary.md,kernel_stats.csv,bench_under_trace.json},
   profiles/<round>_bench_final_{default,siddon}.json, and profiles/traffic.json (the four legs' per-kernel HBM traffic in one table:
   bench.py's `roofline.traffic` and tests/test_bench_contract.py read it)."""
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
rd = sys.argv[1] if len(sys.argv) > 1 else "r05"
out, merged = ROOT / "profiles", {}
for tag in ("trilinear", "pose_only", "siddon", "siddon_nx"):
    src = ROOT / "gpurun_out" / f"{rd}_{tag}"
    for a, b in (("summary.md", "rocprof_summary.md"), ("kernel_stats.csv", "kernel_stats.csv"), ("bench_under_trace.json", "bench_under_trace.json")):
        shutil.copy(src / a, out / f"{rd}_{tag}_{b}")
    for k, v in json.loads((src / "traffic.json").read_text()).items():
        merged.setdefault(k, v)     # (a kernel two legs share keeps the first leg's counters)
(out / "traffic.json").write_text(json.dumps(merged, indent=1))
for f in (f"{rd}_bench_final_default.json", f"{rd}_bench_final_siddon.json", f"{rd}_bench_full_default.json", f"{rd}_bench_full_siddon.json"):
    if (ROOT / "gpurun_out" / f).exists():   # (final = the ONE line the driver parses; full = kernel tables, floors, spreads)
        shutil.copy(ROOT / "gpurun_out" / f, out / f)
print("profiles/ updated for", rd, "-", len(merged), "kernels in traffic.json")

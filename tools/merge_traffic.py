"""Merge the per-run traffic tables that tools/profile.sh writes (gpurun_out/<tag>/traffic.json: FETCH_SIZE + WRITE_SIZE per
launch of every render kernel, from separate rocprofv3 --pmc passes) into profiles/traffic.json, the table bench.py prices
`roofline.traffic` and `hbm_physical` with.     python tools/merge_traffic.py gpurun_out/r03_trilinear gpurun_out/r03_siddon ..."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
out = {}
for d in sys.argv[1:]:
    p = Path(d) / "traffic.json"
    if not p.exists():
        print(f"(no traffic.json under {d})")
        continue
    for k, v in json.loads(p.read_text()).items():
        if k not in out or v.get("dispatches", 0) > out[k].get("dispatches", 0):
            out[k] = v
(ROOT / "profiles" / "traffic.json").write_text(json.dumps(out, indent=1))
print(f"profiles/traffic.json: {len(out)} kernels")

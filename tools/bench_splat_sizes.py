"""The brick-local splat (XVR_DRR_GATHER_SPLAT=1, default) against the fp32 table gather (=0) at smaller volumes / batches:
    python tools/bench_splat_sizes.py        (on the GPU box)"""
import os, subprocess, sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
for size, det, batch in ((128, 128, 116), (256, 256, 116), (256, 256, 16), (512, 256, 8), (384, 256, 116)):
    for mode in ("1", "0"):
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--size", str(size), "--det", str(det), "--batch", str(batch)],
                             env=dict(os.environ, XVR_DRR_GATHER_SPLAT=mode), capture_output=True, text=True)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{size}^3 -> {det}^2, B = {batch}, splat={mode}: step {d['ms_per_step']:.3f} ms, voxel gradient {d['kernels_ms']['trilinear_backward[vol]']:.3f} ms", flush=True)

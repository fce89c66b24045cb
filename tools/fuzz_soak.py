"""Run the seeded fuzz tests of tests/test_hip_parity.py with FRESH seeds (the suite's own are fixed):
    python tools/fuzz_soak.py [first_seed=20000] [count=300]       (on the GPU box)
Prints the seeds that fail with their assertion; exit code = number of failures."""
import sys
import traceback
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import test_hip_parity as T  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
fails = 0
for name, share in (("test_fuzz_random_configurations_against_the_oracle", 1.0), ("test_fuzz_voxel_gather_equals_atomic_scatter", 0.5),
                    ("test_fuzz_drr_module_end_to_end_against_the_oracle", 0.15), ("test_fuzz_fused_similarity_against_torch", 0.15)):
    fn = getattr(T, name)
    n = max(1, int(count * share))
    bad = []
    for seed in range(first, first + n):
        try:
            fn(seed)
        except BaseException as e:   # (pytest.skip raises too)
            if type(e).__name__ == "Skipped":
                continue
            bad.append(seed)
            print(f"{name}[{seed}] FAILED: {type(e).__name__}: {str(e)[:300]}")
            traceback.print_exc(limit=2)
    fails += len(bad)
    print(f"{name}: {n} fresh seeds from {first}, {len(bad)} failed {bad}", flush=True)
sys.exit(min(fails, 100))

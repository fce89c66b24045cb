"""tests/test_fuzz_large.py (the large-launch code paths against the float64 oracle) with FRESH seeds:
    python tools/fuzz_soak_large.py [first_seed=10] [count=30]        (on the GPU box)"""
import sys
import traceback
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import test_fuzz_large as T  # noqa: E402

class _Patch:
    """pytest's monkeypatch.setattr for a direct call: set now, undo() afterwards"""
    def __init__(self):
        self.saved = []

    def setattr(self, obj, name, value):
        self.saved.append((obj, name, getattr(obj, name)))
        setattr(obj, name, value)

    def undo(self):
        for obj, name, value in reversed(self.saved):
            setattr(obj, name, value)
        self.saved = []


first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
fails = 0
for fn in (T.test_large_siddon_launch_against_the_oracle, T.test_large_siddon_launch_under_the_recalled_index_maps,
           T.test_large_trilinear_launch_on_the_tiled_copy_against_the_oracle):
    bad = []
    for seed in range(first, first + count):
        mp = _Patch()
        try:
            fn(seed, mp) if "monkeypatch" in fn.__code__.co_varnames[:fn.__code__.co_argcount] else fn(seed)
        except BaseException as e:  # noqa: BLE001
            bad.append(seed)
            print(f"{fn.__name__}[{seed}] FAILED: {type(e).__name__}: {str(e)[:300]}")
            traceback.print_exc(limit=2)
        finally:
            mp.undo()
    fails += len(bad)
    print(f"{fn.__name__}: {count} fresh seeds from {first}, {len(bad)} failed {bad}", flush=True)
sys.exit(min(fails, 100))

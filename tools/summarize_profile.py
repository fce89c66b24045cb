"""Condense a tools/profile.sh output directory into a small markdown summary (kernel stats + PMC)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
print(f"# rocprofv3 summary ({root.rstrip('/').split('/')[-1]})\n")
import os
stats = sorted(glob.glob(f"{root}/trace/*/*_kernel_stats.csv"), key=os.path.getmtime, reverse=True)   # (a re-used tag: the latest run)
if stats:
    print("## kernel trace (--kernel-trace --stats), whole run incl. warm-up\n")
    print("| kernel | calls | avg ms | total ms | % |\n|---|---|---|---|---|")
    for r in list(csv.DictReader(open(stats[0])))[:14]:
        print(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['TotalDurationNs'])/1e6:.2f} | {r['Percentage']} |")
pm = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
_latest = {}
for f in glob.glob(f"{root}/pmc_*/*/*_counter_collection.csv"):
    d = f.split("/")[-3]
    if d not in _latest or os.path.getmtime(f) > os.path.getmtime(_latest[d]):
        _latest[d] = f
for f in _latest.values():
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("k_trilinear", "k_siddon", "k_gather", "k_backward")):
            continue
        pm[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen and r["Counter_Name"] in ("SQ_WAVES", "FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum"):
            seen.add((k, r["Dispatch_Id"]))
            calls[(k, r["Counter_Name"])] += 1
if pm:
    print("\n## PMC (separate passes; totals over the dispatches of one bench step + the work-count launch)\n")
    print("FETCH_SIZE / WRITE_SIZE are in KiB as reported.  Calibrated in round 4 (profiles/r04_fetch_calibration.txt): one L2 miss\n"
          "is ONE fabric request for a whole 128-byte line, whatever part of it the gather uses (16-byte gathers and coalesced\n"
          "streams alike), and FETCH_SIZE tallies it at 64 bytes: bytes moved = 2 x FETCH_SIZE = 128 x TCC_EA0_RDREQ;\n"
          "TCP_TCC_READ_REQ counts L1 misses in 128-byte lines too.\n")
    for k, v in pm.items():
        n = max(calls[(k, "SQ_WAVES")], calls[(k, "FETCH_SIZE")], calls[(k, "WRITE_SIZE")], 1)
        print(f"### `{k[:100]}`  ({n} dispatches)")
        for c, val in sorted(v.items()):
            print(f"- {c}: {val:.4g}")
        if "SQ_INSTS_VALU" in v and "SQ_THREAD_CYCLES_VALU" in v and v["SQ_INSTS_VALU"]:
            print(f"- derived: active lanes per VALU instruction = {v['SQ_THREAD_CYCLES_VALU']/v['SQ_INSTS_VALU']:.1f} / 64")
        if "TCC_HIT_sum" in v:
            print(f"- derived: L2 hit rate = {v['TCC_HIT_sum']/(v['TCC_HIT_sum']+v['TCC_MISS_sum']):.3f}")
        if "FETCH_SIZE" in v:
            print(f"- derived: FETCH_SIZE per dispatch = {v['FETCH_SIZE']*1024/n/1e9:.3f} GB as reported = {2*v['FETCH_SIZE']*1024/n/1e9:.3f} GB moved (128-byte lines)")
        if "TCC_EA0_RDREQ_sum" in v:
            nn = max(calls[(k, "SQ_WAVES")], calls[(k, "FETCH_SIZE")], 1)
            print(f"- derived: fabric read lines per dispatch = {v['TCC_EA0_RDREQ_sum']/nn:.4g} = {v['TCC_EA0_RDREQ_sum']*128/nn/1e9:.2f} GB")
        if "TCP_TCC_READ_REQ_sum" in v:
            nn = max(calls[(k, "SQ_WAVES")], calls[(k, "FETCH_SIZE")], 1)
            print(f"- derived: L1-miss lines per dispatch = {v['TCP_TCC_READ_REQ_sum']/nn:.4g} = {v['TCP_TCC_READ_REQ_sum']*128/nn/1e9:.2f} GB"
                  + (f", mean latency {v['TCP_TCC_READ_REQ_LATENCY_sum']/v['TCP_TCC_READ_REQ_sum']:.0f} clk" if "TCP_TCC_READ_REQ_LATENCY_sum" in v and v["TCP_TCC_READ_REQ_sum"] else ""))
        if "WRITE_SIZE" in v:
            print(f"- derived: WRITE_SIZE per dispatch = {v['WRITE_SIZE']*1024/n/1e9:.3f} GB")
        print()
import json  # noqa: E402
traffic = {}
for k, v in pm.items():
    n = max(calls[(k, "FETCH_SIZE")], calls[(k, "WRITE_SIZE")], 1)
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
        # fetch_bytes: the bytes MOVED -- twice the reported FETCH_SIZE (128-byte lines tallied at 64: profiles/r04_fetch_calibration.txt)
        traffic[k] = {"fetch_bytes": 2.0 * v.get("FETCH_SIZE", 0.0) * 1024 / n, "fetch_bytes_reported": v.get("FETCH_SIZE", 0.0) * 1024 / n,
                      "write_bytes": v.get("WRITE_SIZE", 0.0) * 1024 / n, "dispatches": n}
with open(f"{root}/traffic.json", "w") as f:
    json.dump(traffic, f, indent=1)

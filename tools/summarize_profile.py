#!/usr/bin/env python3
"""Turns a tools/profile.sh output directory (gpurun_out/<tag>) into tracked files under profiles/."""
import collections
import csv
import os
import shutil
import sys

src, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "stats", "r_kernel_stats.csv"), os.path.join(dst, name + "_kernel_stats.csv"))


def pmc(kind):
    path = os.path.join(src, kind, "r_counter_collection.csv")
    agg = collections.defaultdict(lambda: [0, 0.0])
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            agg[r["Kernel_Name"]][0] += 1
            agg[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return agg


fetch, write = pmc("pmc_fetch"), pmc("pmc_write")
stats = list(csv.DictReader(open(os.path.join(src, "stats", "r_kernel_stats.csv"))))
bench = ""
for line in open(os.path.join(src, "stats.log"), errors="ignore"):
    if line.startswith('{"metric"'):
        bench = line.strip()
with open(os.path.join(dst, name + "_summary.md"), "w") as f:
    f.write("# %s — rocprofv3 summary\n\n" % name)
    f.write("Command: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 40 --warmup 4 --cpu-baseline-frames 0` "
            "(+ separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs of the same command; tools/profile.sh).\n\n")
    f.write("FETCH_SIZE / WRITE_SIZE are KiB per launch as reported; per MI355X_MICROARCH.md §HBM, FETCH_SIZE on gfx950 "
            "reports exactly half of the bytes of wide coalesced reads, so `fetch x2` is the corrected figure for streaming kernels.\n\n")
    f.write("| kernel | calls | avg us | min us | max us | % | FETCH_SIZE KiB/launch | fetch x2 MB | WRITE_SIZE KiB/launch |\n|---|---|---|---|---|---|---|---|---|\n")
    for r in stats:
        k = r["Name"]
        if not ("chord::" in k or "rocclr" in k):
            continue
        fe = fetch.get(k, [0, 0.0]); wr = write.get(k, [0, 0.0])
        fa = fe[1] / fe[0] if fe[0] else float("nan")
        wa = wr[1] / wr[0] if wr[0] else float("nan")
        f.write("| `%s` | %s | %.1f | %.1f | %.1f | %s | %.0f | %.1f | %.0f |\n" % (
            k.replace("void ", "").split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
            float(r["MaxNs"]) / 1e3, r["Percentage"], fa, fa * 2 * 1024 / 1e6, wa))
    if bench:
        f.write("\nbench.py line of the `--stats` run (timing perturbed by the profiler):\n\n```json\n%s\n```\n" % bench)
# per-kernel HBM traffic for bench.py's roofline.traffic: 2 x FETCH_SIZE (gfx950 correction for wide reads,
# MI355X_MICROARCH.md "HBM") + WRITE_SIZE, KiB -> bytes, per launch
import json
traffic = {}
for r in stats:
    k = r["Name"]
    if "chord::" not in k:
        continue
    fe = fetch.get(k, [0, 0.0]); wr = write.get(k, [0, 0.0])
    if not fe[0] or not wr[0]:
        continue
    short = k.replace("void ", "").replace("chord::", "").split("(")[0].split("<")[0]
    t = traffic.setdefault(short, {"fetch_kib_per_launch": 0.0, "write_kib_per_launch": 0.0, "launches": 0, "avg_us": 0.0})
    calls = int(r["Calls"])
    # template instances of one kernel are merged, weighted by their launches
    t["fetch_kib_per_launch"] = (t["fetch_kib_per_launch"] * t["launches"] + fe[1] / fe[0] * calls) / (t["launches"] + calls)
    t["write_kib_per_launch"] = (t["write_kib_per_launch"] * t["launches"] + wr[1] / wr[0] * calls) / (t["launches"] + calls)
    t["avg_us"] = (t["avg_us"] * t["launches"] + float(r["AverageNs"]) / 1e3 * calls) / (t["launches"] + calls)
    t["launches"] += calls
for t in traffic.values():
    t["hbm_bytes_per_launch"] = int((2.0 * t["fetch_kib_per_launch"] + t["write_kib_per_launch"]) * 1024)
try:
    import subprocess
    head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    head = ""
json.dump({"profile_head": head,       # the commit the tree stood at when the profile was summarised (bench.py reports it beside roofline.traffic)
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python bench.py --steps 40 --warmup 4` (tools/profile.sh); "
                     "bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 per launch", "kernels": traffic},
          open(os.path.join(dst, name + "_traffic.json"), "w"), indent=1)
print("wrote", os.path.join(dst, name + "_summary.md"))

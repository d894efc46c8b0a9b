#!/usr/bin/env python3
"""The tile kernel's SQ counters PER PASS: the per-dispatch rows of the counter passes of a round (gpurun_out/rNN_sq*/r_counter_collection.csv)
grouped by the launch's grid instead of by kernel name -- a first pass under the schedule the frame before made has one workgroup more
(2 041 x 512 threads at 4K) than a direct second pass or a pass under a schedule kernel's schedule (2 040).

  python tools/counters_by_pass.py r06 > profiles/r06_config3_tile_kernel_by_pass.md
"""
import collections
import csv
import glob
import os
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r06"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", R + "_sq[0-9]", "r_counter_collection.csv"))):
    seen = set()
    for r in csv.DictReader(open(path)):
        if "raster_tile_kernel" not in r["Kernel_Name"]:
            continue
        g = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
        agg[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# %s config 3 (street_4k_hzb): `raster_tile_kernel` counters per launch, grouped by the launch's workgroups\n" % R)
print("Source: the per-dispatch rows of the six `rocprofv3 --pmc` passes of `tools/round_profile.sh %s` (`bench.py --steps 12 --warmup 4`; tools/counters_by_pass.py)." % R)
print("2 041 workgroups = a first pass under the schedule the same pass of the frame before made (one workgroup more: it makes the next one);")
print("2 040 = a direct second pass, or a first pass under a schedule kernel's schedule (the bench's moving-path run with fresh schedules, a context's first frame).\n")
names = sorted({c for g in agg for c in agg[g]})
first = max((g for g in agg if len(dur[g]) >= 8), default=None)          # the first passes under a kept schedule: the larger grid
if first is not None and "SQ_BUSY_CYCLES" in agg[first] and "SQ_INSTS_VALU" in agg[first]:
    import json
    import subprocess
    m = {c: sum(x) / len(x) for c, x in agg[first].items()}
    cyc = m["SQ_BUSY_CYCLES"] / 32.0
    head = os.environ.get("PROFILE_HEAD") or subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()   # (the commit the counter passes ran at)
    json.dump({"kernel": "raster_tile_kernel, first pass of a config-3 frame (launches of %d workgroups)" % first, "workload": "street_4k_hzb",
               "valu_insts_per_launch": round(m["SQ_INSTS_VALU"]), "salu_insts_per_launch": round(m.get("SQ_INSTS_SALU", 0)),
               "busy_cycles_per_launch": round(cyc), "clock_ghz_from_busy_cycles": round(cyc / (sum(dur[first]) / len(dur[first])) / 1e3, 3),
               "mean_launch_us": round(sum(dur[first]) / len(dur[first]), 2), "simds": 1024, "cycles_per_wave64_instruction": 4,
               "valu_busy_frac": round(m["SQ_INSTS_VALU"] / 1024 * 4 / cyc, 4),
               "waves_resident_per_simd": round(m.get("SQ_WAVE_CYCLES", 0) * 4 / 1024 / cyc, 2),
               "lanes_active_frac": round(m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_ACTIVE_INST_VALU"] * 64), 4) if "SQ_THREAD_CYCLES_VALU" in m and m.get("SQ_ACTIVE_INST_VALU") else None,
               "profile_head": head}, open(os.path.join(ROOT, "profiles", "%s_config3_tile_valu.json" % R), "w"), indent=1)
for g in sorted(agg, reverse=True):
    if len(dur[g]) < 8:
        continue
    v = {c: sum(x) / len(x) for c, x in agg[g].items()}
    d = sum(dur[g]) / len(dur[g])
    print("## %d workgroups: %d launches over the passes, mean duration %.1f us\n" % (g, len(dur[g]), d))
    print("| counter | mean per launch |\n|---|---|")
    for c in names:
        if c in v:
            print("| %s | %.0f |" % (c, v[c]))
    if "SQ_BUSY_CYCLES" in v and "SQ_INSTS_VALU" in v:
        cyc = v["SQ_BUSY_CYCLES"] / 32.0
        print("\nSQ_BUSY_CYCLES / 32 shader engines = %.0f cycles per launch = %.2f GHz over %.1f us; SQ_INSTS_VALU / 1 024 SIMDs x 4 cycles = %.0f issue cycles per SIMD = **%.0f %% of the launch**;"
              % (cyc, cyc / d / 1e3, d, v["SQ_INSTS_VALU"] / 1024 * 4, 100 * v["SQ_INSTS_VALU"] / 1024 * 4 / cyc))
        if "SQ_WAVE_CYCLES" in v:
            print("SQ_WAVE_CYCLES x 4 / 1 024 SIMDs / those cycles = %.2f waves resident per SIMD (of 4)." % (v["SQ_WAVE_CYCLES"] * 4 / 1024 / cyc))
    print()

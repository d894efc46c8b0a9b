#!/usr/bin/env python3
"""Turns the gpurun_out/ files of one `tools/round_profile.sh rNN` call into the tracked evidence under profiles/:

  python tools/collect_round.py r05

  rNN_config3_4k_hzb_* / config4_x64_4k_hzb_* / config5_subpixel_1g_* / config5_hotspot_* / masked_4k_*   tools/summarize_profile.py
  rNN_config3_4k_hzb_sq_counters.md, rNN_config5_sq_counters.md      per-kernel averages of the SQ counter passes (tools/pmc.sh) + ratios
  rNN_bench_lines.txt                                                one compact line per bench.py run of the call
  rNN_shard_time_*.txt, rNN_config{3,4}_timeline.txt, rNN_rank3_kernel_stats.csv, rNN_shadow_time.txt, rNN_group_host_time.txt   copied
  rNN_kernel_resources.md                                            tools/kernel_resources.py --md (from the compiler, no GPU)
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r05"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")

for tag, name in (("c3", "config3_4k_hzb"), ("c4", "config4_x64_4k_hzb"), ("c5", "config5_subpixel_1g"), ("c5hot", "config5_hotspot"), ("masked", "masked_4k")):
    src = os.path.join(G, "%s_%s" % (R, tag))
    if os.path.isdir(os.path.join(src, "stats")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_profile.py"), src, "%s_%s" % (R, name)])


def counters(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for d in dirs:
        path = os.path.join(G, d, "r_counter_collection.csv")
        if not os.path.exists(path):
            continue
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].replace("void ", "").replace("chord::", "").split("(")[0]
            a = agg[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: {c: v[1] / v[0] for c, v in cs.items()} for k, cs in agg.items()}


def sq_table(dirs, out, title, note):
    data = counters(dirs)
    cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
            "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC",
            "SQ_INSTS_BRANCH", "SQ_IFETCH", "SQ_IFETCH_LEVEL", "SQ_THREAD_CYCLES_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL",
            "SQ_INSTS_LDS_ATOMIC", "SQ_INSTS_LDS_LOAD", "SQ_INSTS_LDS_STORE", "SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_SMEM",
            "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_VMEM_WR_TA_DATA_FIFO_FULL", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_BUSY_CU_CYCLES"]
    cols = [c for c in cols if any(c in v for v in data.values())]
    if not data:                                   # (a partial call: no counter passes for this table -- the committed one stays)
        return data
    with open(os.path.join(P, out), "w") as f:
        f.write("# %s\n\n%s\n\n" % (title, note))
        f.write("| kernel | " + " | ".join(c.replace("SQ_", "") for c in cols) + " | wait / wave-cycles | active inst / wave-cycles | VALU lanes active | LDS bank conflicts |\n")
        f.write("|---|" + "---|" * (len(cols) + 4) + "\n")
        for k in sorted(data):
            v = data[k]
            if not any(s in k for s in ("raster", "cull", "hzb", "detile")):
                continue

            def ratio(a, b, scale=1.0):
                return "%.1f %%" % (100.0 * v[a] / (v[b] * scale)) if a in v and b in v and v[b] else "-"
            f.write("| `%s` | " % k + " | ".join("%d" % round(v[c]) if c in v else "-" for c in cols)
                    + " | %s | %s | %s | %s |\n" % (ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), ratio("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"),
                                                     ratio("SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", 64.0), ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")))
        f.write("\n(per launch, summed over the chip as rocprofv3 reports them; VALU lanes active = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64).)\n")
    return data


c3 = sq_table(["%s_sq1" % R, "%s_sq2" % R, "%s_sq3" % R, "%s_sq4" % R, "%s_sq5" % R, "%s_sq6" % R], "%s_config3_4k_hzb_sq_counters.md" % R,
              "%s config 3 (street_4k_hzb) -- SQ counters per kernel launch" % R,
              "Averages over `bench.py --steps 12 --warmup 4`, three separate `rocprofv3 --pmc` passes (tools/pmc.sh), the round's final kernels.")
c5 = sq_table(["%s_sq1_c5" % R, "%s_sq2_c5" % R], "%s_config5_sq_counters.md" % R,
              "%s block kernel -- SQ counters per launch on `subpixel_64m --debug-flags 65536` (config 5 at 1/16 size, block kernel forced: 524 288 clusters per launch)" % R,
              "Two separate `rocprofv3 --pmc` passes (tools/pmc.sh).")
blk = next((v for k, v in c5.items() if "raster_setup_blocks_kernel" in k), None)
if blk and "SQ_INSTS_VALU" in blk:
    with open(os.path.join(P, "%s_config5_sq_counters.md" % R), "a") as f:
        f.write("\nPer cluster (524 288 per launch): **%.0f VALU + %.0f SALU** wave-instructions.\n" % (blk["SQ_INSTS_VALU"] / 524288.0, blk.get("SQ_INSTS_SALU", 0.0) / 524288.0))
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        head = ""
    # what bench.py's roofline.valu reads (like *_traffic.json: a counter pass cannot run inside the bench process)
    json.dump({"kernel": "raster_setup_blocks_kernel", "workload": "subpixel_64m --debug-flags 65536", "clusters_per_launch": 524288,
               "valu_per_cluster": round(blk["SQ_INSTS_VALU"] / 524288.0, 1), "salu_per_cluster": round(blk.get("SQ_INSTS_SALU", 0.0) / 524288.0, 1),
               "profile_head": head}, open(os.path.join(P, "%s_config5_valu.json" % R), "w"), indent=1)

# bench lines, compact
with open(os.path.join(P, "%s_bench_lines.txt" % R), "w") as f:
    f.write("bench.py lines of the round-profile call (one box; 200 steps unless the workload says otherwise): ms per step | Gtri/s | GPU stamps (us per frame): cull, setup, clip+order, tile | launches per frame\n")
    for path in sorted(glob.glob(os.path.join(G, "%s_bench_*.json" % R))):
        try:
            d = json.load(open(path))
        except ValueError:
            f.write("%-40s (no line)\n" % os.path.basename(path))
            continue
        g = d["gpu_ms"]
        f.write("%-40s %-24s steps %3d  %.4f ms  %8.3f Gtri/s  cull %6.1f setup %7.1f clip+order %5.1f tile %7.1f  launches %s  roofline %s frac %.4f\n" % (
            os.path.basename(path).replace("%s_bench_" % R, "").replace(".json", ""), d["config"]["workload"], d["steps"], d["ms_per_step"], d["value"],
            g["msInstanceCulling"] * 1e3, g["msRasterCluster"] * 1e3, g["msRasterClip"] * 1e3, g["msRasterChunk"] * 1e3, d.get("kernel_launches"),
            d["roofline"]["kernel"].split(" ")[0], d["roofline"]["frac"]))
    for name in ("default", "default_20steps"):
        path = os.path.join(G, "%s_bench_%s.json" % (R, name))
        if os.path.exists(path):
            f.write("\n%s (full line):\n%s\n" % (name, open(path).read().strip()))

for src, dst in (("%s_shard_time_c5.txt", "%s_shard_time_config5.txt"), ("%s_shard_time_c4.txt", "%s_shard_time_config4.txt"),
                 ("%s_shard_time_c4_pipelined.txt", "%s_shard_time_config4_pipelined.txt"), ("%s_shard_time_c5hot.txt", "%s_shard_time_config5_hotspot.txt"),
                 ("%s_timeline.txt", "%s_config3_timeline.txt"), ("%s_timeline_c4.txt", "%s_config4_timeline.txt"), ("%s_timeline_rank3_c4.txt", "%s_config4_rank3_of_8_timeline.txt"),
                 ("%s_microbench_valu_issue.txt", "%s_microbench_valu_issue.txt"), ("%s_tile_profile.txt", "%s_tile_profile_config3.txt"), ("%s_ablate_tile.txt", "%s_tile_kernel_ablation.txt"),
                 ("%s_shadow_time.txt", "%s_shadow_time.txt"), ("%s_group_host_time.txt", "%s_group_host_time.txt")):
    s = os.path.join(G, src % R)
    if os.path.exists(s):
        shutil.copy(s, os.path.join(P, dst % R))
for s in glob.glob(os.path.join(G, "%s_rank3" % R, "**", "r_kernel_stats.csv"), recursive=True):
    shutil.copy(s, os.path.join(P, "%s_rank3_of_8_config5_kernel_stats.csv" % R))
md = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "--md"], capture_output=True, text=True).stdout
open(os.path.join(P, "%s_kernel_resources.md" % R), "w").write(md)
print("collected", R)

# the tile kernel's counters per pass (first passes have one workgroup more: tools/counters_by_pass.py)
with open(os.path.join(P, "%s_config3_tile_kernel_by_pass.md" % R), "w") as f:
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "counters_by_pass.py"), R], stdout=f)

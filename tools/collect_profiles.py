#!/usr/bin/env python3
"""Copies the judged summaries of a tools/gpu/profile_bench_and_pmc.sh run from gpurun_out/
(scratch) into profiles/ (tracked): bench JSON lines, rocprofv3 kernel stats, PMC summary with
the gfx950 FETCH_SIZE x2 correction (MI355X_MICROARCH.md section HBM), and pmc_traffic.json
(read by bench.py for roofline.traffic)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"

shutil.copy(os.path.join(G, "bench.json"), os.path.join(P, f"{tag}_bench_u32w7.json"))
shutil.copy(os.path.join(G, "bench_other.jsonl"), os.path.join(P, f"{tag}_bench_other_workloads.jsonl"))
shutil.copy(os.path.join(G, "prof_r01_trace", "bench_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_u32w7_kernel_stats.csv"))
with open(os.path.join(G, "prof_bench.log")) as f, open(os.path.join(P, f"{tag}_bench_u32w7_under_rocprof.json"), "w") as o:
    o.writelines(l for l in f if l.startswith('{"metric"'))

rows = {}
for which in ("fetch", "write"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(G, f"prof_r01_{which}", "pmc_counter_collection.csv"))):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    rows[which] = {k: sum(v) / len(v) for k, v in agg.items() if max(v) > 1e5 and "distribution" not in k}
lines = ["kernel,launches_avg_over,FETCH_SIZE_KB_raw,FETCH_bytes_corrected_x2,WRITE_SIZE_KB_raw,WRITE_bytes,hbm_bytes_per_launch"]
traffic = {}
for k in sorted(set(rows["fetch"]) | set(rows["write"])):
    f, w = rows["fetch"].get(k, 0.0), rows["write"].get(k, 0.0)
    fb, wb = f * 1024 * 2, w * 1024
    lines.append(f'"{k}",3,{f:.0f},{fb:.0f},{w:.0f},{wb:.0f},{fb + wb:.0f}')
    traffic[k] = int(fb + wb)
open(os.path.join(P, f"{tag}_pmc_fetch_write_summary.csv"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
names = {"u32_w7_unpack": "k_unpack<unsigned int, 7, 0>", "u32_w7_pack": "k_pack<unsigned int, 7, 0>",
         "u32_w12_undelta_pack": "k_unpack<unsigned int, 12, 2>", "u64_w17_unpack": "k_unpack<unsigned long, 17, 0>",
         "u64_w17_pack": "k_pack<unsigned long, 17, 0>"}
pj = os.path.join(P, "pmc_traffic.json")
j = json.load(open(pj))
for wl, pat in names.items():
    for k, v in traffic.items():
        if pat in k:
            j[wl] = {"hbm_bytes_per_launch_at_10M_blocks": v}
j["_source"] = (f"profiles/{tag}_pmc_fetch_write_summary.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, "
                "tools/pmc_probe.py, 10 M blocks). FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM, confirmed in the same "
                "run on an 8 GiB copyBuffer (FETCH raw = exactly half of 8 GiB; WRITE raw = 8 GiB, factor 1).")
json.dump(j, open(pj, "w"), indent=1)
for l in open(os.path.join(G, "bench_other.jsonl")):
    d = json.loads(l)
    print(d["config"]["workload"][:44], d["value"], "Gint/s", d["roofline"]["achieved"], "GB/s", d["roofline"]["frac"])
d = json.load(open(os.path.join(G, "bench.json")))
print("HEADLINE", d["value"], d["roofline"]["achieved"], d["roofline"]["frac"], d["cpu_baseline"]["value"])

#!/usr/bin/env python3
"""Copies the judged summaries of a `bash tools/gpu/evidence.sh <round>` run from gpurun_out/<round>/ (scratch) into
profiles/ (tracked), named <round>_*.      python tools/collect_profiles.py r03"""
import glob
import json
import os
import shutil
import sys

RD = sys.argv[1] if len(sys.argv) > 1 else "r06"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", RD)
P = os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, os.path.join(P, dst))
        print("copied", dst)
    else:
        print("MISSING", src)


cp(os.path.join(G, "bench_u32w7.json"), f"{RD}_bench_u32w7.json")
cp(os.path.join(G, "bench_other.jsonl"), f"{RD}_bench_other_workloads.jsonl")
cp(os.path.join(G, "host_latency.txt"), f"{RD}_host_latency.txt")
cp(os.path.join(G, "pmc_unpack_single.txt"), f"{RD}_pmc_unpack_single.txt")
cp(os.path.join(G, "sq_mixed", "sq_derived.txt"), f"{RD}_sq_mixed_final.txt")
cp(os.path.join(G, "full_check.txt"), f"{RD}_full_check.txt")
cp(os.path.join(G, "full_check_badscan.txt"), f"{RD}_full_check_known_bad_build.txt")
for src, dst in (("bench_8ranks_one_device.json", f"{RD}_bench_8ranks_one_device_gloo.json"),
                 ("bench_2ranks_gloo.json", f"{RD}_bench_2ranks_one_device_gloo.json"),
                 ("bench_2ranks_auto.json", f"{RD}_bench_2ranks_one_device_rccl_attempt.json")):
    if os.path.exists(os.path.join(G, src)):
        with open(os.path.join(G, src)) as f, open(os.path.join(P, dst), "w") as o:
            o.writelines(l for l in f if l.startswith("{"))
        print("copied", dst)
cp(os.path.join(G, "multi_gpu_decode.json"), f"{RD}_multi_gpu_decode_c_driver.json")
cp(os.path.join(G, "multi_gpu_decode_2threads.json"), f"{RD}_multi_gpu_decode_c_driver_2threads.json")
cp(os.path.join(G, "ablayouts.txt"), f"{RD}_ablayouts.txt")
for c in ("quick", "quick_constructed", "fused", "consume", "refbench", "batch", "mixed", "mixed_separate", "allwidths", "allwidths_constructed", "single"):
    cp(os.path.join(G, f"sweep_{c}.txt"), f"{RD}_sweep_{c}.txt")
# one rocprofv3 kernel trace per BASELINE config (2, 5, 3 unpack, 3 pack, 4): the stats summary + the line printed inside that run
for tag, dst in (("prof_trace", f"{RD}_bench_u32w7"), ("prof_trace_mixed", f"{RD}_bench_u32_mixed"), ("prof_trace_u64_unpack", f"{RD}_bench_u64w17_unpack"),
                 ("prof_trace_u64_pack", f"{RD}_bench_u64w17_pack"), ("prof_trace_undelta_pack", f"{RD}_bench_u32w12_undelta_pack")):
    for f in glob.glob(os.path.join(G, tag, "**", "*kernel_stats.csv"), recursive=True):
        cp(f, dst + "_kernel_stats.csv")
    src = os.path.join(G, tag + ".log")
    if os.path.exists(src):
        with open(src) as f, open(os.path.join(P, dst + "_under_rocprof.json"), "w") as o:
            o.writelines(l for l in f if l.startswith('{"metric"'))
        print("copied", dst + "_under_rocprof.json")
# SQ / LDS counter passes (bash tools/gpu/sq_counters.sh tools/pmc_probe_r03.py gpurun_out/<round>), when the run made them
if os.path.exists(os.path.join(G, "sq_counters.csv")):
    cp(os.path.join(G, "sq_counters.csv"), f"{RD}_pmc_sq_counters.csv")
    with open(os.path.join(G, "sq_derived.txt")) as f, open(os.path.join(P, f"{RD}_pmc_sq_derived.txt"), "w") as o:
        o.write(f"# from profiles/{RD}_pmc_sq_counters.csv + the kernel trace of the same rocprofv3 pass (tools/gpu/sq_counters.sh "
                "tools/pmc_probe_r03.py), the round's final kernels\n")
        o.write(f.read())
    print("copied", f"{RD}_pmc_sq_derived.txt")
try:
    d = json.load(open(os.path.join(G, "bench_u32w7.json")))
    print("HEADLINE", d["value"], d["roofline"]["achieved"], d["roofline"]["frac"], "traffic", d["roofline"]["traffic"],
          "cpu", d["cpu_baseline"]["value"], "config5", d["config5_strong"]["value"], d["config5_strong"]["per_rank"][0]["frac"])
except Exception as e:
    print("no headline:", e)
for l in open(os.path.join(G, "bench_other.jsonl")):
    d = json.loads(l)
    print(d["config"]["workload"][:60], d["value"], "Gint/s", d["roofline"]["achieved"], "GB/s", d["roofline"]["frac"],
          "traffic", d["roofline"]["traffic"], "cpu", (d.get("cpu_baseline") or {}).get("value"))

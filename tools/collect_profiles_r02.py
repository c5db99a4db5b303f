#!/usr/bin/env python3
"""Copies the judged summaries of a tools/gpu/r02_evidence.sh run from gpurun_out/r02/ (scratch) into profiles/ (tracked)."""
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "r02")
P = os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, dst))
        print("copied", dst)
    else:
        print("MISSING", src)


cp(os.path.join(G, "bench_u32w7.json"), "r02_bench_u32w7.json")
cp(os.path.join(G, "bench_other.jsonl"), "r02_bench_other_workloads.jsonl")
cp(os.path.join(G, "host_latency.txt"), "r02_host_latency.txt")
for c in ("quick", "fused", "consume"):
    cp(os.path.join(G, f"sweep_{c}.txt"), f"r02_sweep_{c}.txt")
for tag, dst in (("prof_trace", "r02_bench_u32w7"), ("prof_trace_mixed", "r02_bench_u32_mixed")):
    for f in glob.glob(os.path.join(G, tag, "**", "*kernel_stats.csv"), recursive=True):
        cp(f, dst + "_kernel_stats.csv")
for log, dst in (("bench_under_rocprof.log", "r02_bench_u32w7_under_rocprof.json"),
                 ("bench_mixed_under_rocprof.log", "r02_bench_u32_mixed_under_rocprof.json")):
    src = os.path.join(G, log)
    if os.path.exists(src):
        with open(src) as f, open(os.path.join(P, dst), "w") as o:
            o.writelines(l for l in f if l.startswith('{"metric"'))
for name in ("bench_u32w7.json",):
    try:
        d = json.load(open(os.path.join(G, name)))
        print("HEADLINE", d["value"], d["roofline"]["achieved"], d["roofline"]["frac"], "traffic", d["roofline"]["traffic"],
              "cpu", d["cpu_baseline"]["value"], "config5", d["config5_strong"]["value"], d["config5_strong"]["per_rank"][0]["frac"])
    except Exception as e:
        print("no headline:", e)
for l in open(os.path.join(G, "bench_other.jsonl")):
    d = json.loads(l)
    print(d["config"]["workload"][:60], d["value"], "Gint/s", d["roofline"]["achieved"], "GB/s", d["roofline"]["frac"],
          "traffic", d["roofline"]["traffic"], "cpu", (d.get("cpu_baseline") or {}).get("value"))

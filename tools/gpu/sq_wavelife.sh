# tools/gpu/sq_wavelife.sh <probe.py> <outdir>: where does a wavefront's life go?  Wave cycles, waiting, issue by category (own --pmc passes,
# --kernel-trace only), per kernel: averages over the launches + per-wave figures.
PROBE=$1
R=$2
mkdir -p $R
ROOT=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rm -rf $ROOT/$R/wl1 $ROOT/$R/wl2 && \
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $ROOT/$R/wl1 -o wl1 -- python $ROOT/$PROBE > $ROOT/$R/wl1.log 2>&1; echo "rc=$?"; \
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $ROOT/$R/wl2 -o wl2 -- python $ROOT/$PROBE > $ROOT/$R/wl2.log 2>&1; echo "rc=$?" )
tail -2 $R/wl2.log
OUTDIR=$R python - <<'PY'
import collections, csv, glob, os
R = os.environ["OUTDIR"]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in ("wl1", "wl2"):
    for f in glob.glob(f"{R}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fl::k_" in r["Kernel_Name"] and "k_scan" not in r["Kernel_Name"]:
                rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{R}/wl1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fl::k_" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(f"{R}/wavelife.txt", "w") as o:
    o.write("kernel | median us | waves | per wave: cycles alive, cycles waiting (any), cycles waiting to issue, VALU / SALU / SMEM / branch / VMEM rd / VMEM wr / LDS instructions | "
            "waves resident per SIMD = wave cycles / (busy cycles * 4)... (SQ_BUSY_CYCLES is per SE: see the raw ratio)\n")
    for k in sorted(rows):
        v = {n: (sum(x) / len(x)) for n, x in rows[k].items()}
        w = max(v.get("SQ_WAVES", 1), 1)
        t = sorted(dur[k])[len(dur[k]) // 2] if k in dur else 0
        g = lambda n: v.get(n, 0.0) / w
        o.write("%-70s | %8.1f | %9.0f | alive %7.0f  wait_any %7.0f  wait_inst %7.0f | valu %5.0f salu %5.0f smem %4.0f branch %4.0f vmem_rd %4.1f vmem_wr %4.1f lds %4.1f | wave_cycles/busy_cycles %.1f\n" % (
            k.replace("void fl::", "").replace("(fl::", "(")[:70], t / 1e3, w, g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY"), g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"),
            g("SQ_INSTS_SMEM"), g("SQ_INSTS_BRANCH"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"), g("SQ_INSTS_LDS"), v.get("SQ_WAVE_CYCLES", 0) / max(v.get("SQ_BUSY_CYCLES", 1), 1)))
print(open(f"{R}/wavelife.txt").read())
PY

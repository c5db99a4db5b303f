# SQ / LDS counters of whatever kernels a probe script launches: where do wave cycles go, do the LDS images conflict?
#   bash tools/gpu/sq_counters.sh tools/pmc_probe_r03.py gpurun_out/r03      (on the GPU box; writes <outdir>/sq_counters.csv)
# Counters in their own passes with --kernel-trace only (no other trace domain), as MI355X_MICROARCH.md prescribes.
PROBE=$1
R=$2
mkdir -p $R
ROOT=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rm -rf $ROOT/$R/prof_sq $ROOT/$R/prof_sq2 && \
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $ROOT/$R/prof_sq -o sq -- python $ROOT/$PROBE > $ROOT/$R/prof_sq.log 2>&1; echo "rc=$?"; \
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $ROOT/$R/prof_sq2 -o sq2 -- python $ROOT/$PROBE > $ROOT/$R/prof_sq2.log 2>&1; echo "rc=$?" )
tail -2 $R/prof_sq2.log
OUTDIR=$R python - <<'PY'
import collections, csv, glob, os
R = os.environ["OUTDIR"]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in ("prof_sq", "prof_sq2"):
    for f in glob.glob(f"{R}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fl::k_" in r["Kernel_Name"] and "k_scan" not in r["Kernel_Name"]:
                rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
         "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VMEM"]
with open(f"{R}/sq_counters.csv", "w") as o:
    o.write("kernel," + ",".join(names) + ",valu_active_frac_of_wave_cycles,valu_active_frac_of_busy_cycles_x4simd,lds_conflict_frac_of_lds_active\n")
    for k in sorted(rows):
        v = {n: (sum(rows[k][n]) / len(rows[k][n]) if rows[k][n] else 0.0) for n in names}
        o.write('"%s",' % k + ",".join("%.0f" % v[n] for n in names) +
                ",%.3f,%.3f,%.3f\n" % (v["SQ_ACTIVE_INST_VALU"] / max(v["SQ_WAVE_CYCLES"], 1), v["SQ_ACTIVE_INST_VALU"] / max(4 * v["SQ_BUSY_CYCLES"], 1),
                                 v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1)))
print(open(f"{R}/sq_counters.csv").read())
# derived: how busy are the VALUs / the LDS over the kernel's wall time?  A wave64 VALU instruction occupies its SIMD for 4 cycles;
# the chip has 256 CUs x 4 SIMDs; 2.4 GHz is the nominal clock (lower under load, so the true fractions are a little higher)
dur = collections.defaultdict(list)
for f in glob.glob(f"{R}/prof_sq/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fl::k_" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(f"{R}/sq_derived.txt", "w") as o:
    o.write("kernel | median us | VALU issue utilisation = SQ_INSTS_VALU*4 / (t * 2.4 GHz * 1024 SIMDs) | VALU insts per wave | "
            "LDS busy = SQ_LDS_IDX_ACTIVE / (t * 2.4 GHz * 256 CUs) | of which bank conflicts\n")
    for k in sorted(rows):
        if k not in dur:
            continue
        v = {n: (sum(rows[k][n]) / len(rows[k][n]) if rows[k][n] else 0.0) for n in names}
        t = sorted(dur[k])[len(dur[k]) // 2]
        cyc = t * 2.4
        o.write("%-74s | %8.1f | %.2f | %5.0f | %.2f | %.2f\n" % (k.replace("void fl::", "").replace("(fl::", "(")[:74], t / 1e3,
                v["SQ_INSTS_VALU"] * 4 / (cyc * 1024), v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1), v["SQ_LDS_IDX_ACTIVE"] / (cyc * 256),
                v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1)))
print(open(f"{R}/sq_derived.txt").read())
PY

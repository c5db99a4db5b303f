# SQ / LDS counters of the round-2 (wave-per-block) kernels: where do wave cycles go, do the LDS images conflict?
R=gpurun_out/r02
mkdir -p $R
ROOT=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rm -rf $ROOT/$R/prof_sq $ROOT/$R/prof_sq2 && \
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $ROOT/$R/prof_sq -o sq -- python $ROOT/tools/pmc_probe_r02.py > $ROOT/$R/prof_sq.log 2>&1; echo "rc=$?"; \
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $ROOT/$R/prof_sq2 -o sq2 -- python $ROOT/tools/pmc_probe_r02.py > $ROOT/$R/prof_sq2.log 2>&1; echo "rc=$?" )
tail -2 $R/prof_sq2.log
python - <<'PY'
import collections, csv, glob
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("prof_sq", "prof_sq2"):
    for f in glob.glob(f"gpurun_out/r02/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fl::k_" in r["Kernel_Name"] and "k_scan" not in r["Kernel_Name"]:
                rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
         "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VMEM"]
with open("gpurun_out/r02/sq_counters.csv", "w") as o:
    o.write("kernel," + ",".join(names) + ",valu_active_frac_of_wave_cycles,lds_conflict_frac_of_lds_active\n")
    for k in sorted(rows):
        v = {n: (sum(rows[k][n]) / len(rows[k][n]) if rows[k][n] else 0.0) for n in names}
        o.write('"%s",' % k + ",".join("%.0f" % v[n] for n in names) +
                ",%.3f,%.3f\n" % (v["SQ_ACTIVE_INST_VALU"] / max(v["SQ_WAVE_CYCLES"], 1), v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1)))
print(open("gpurun_out/r02/sq_counters.csv").read())
PY

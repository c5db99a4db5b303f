# tools/gpu/scatter_calib.sh <outdir>: scattered reads of 4 .. 128 bytes per random line -- time per access, and FETCH_SIZE / EA request counters per access
R=${1:-gpurun_out/r06b/scatter}
mkdir -p $R
ROOT=$(pwd)
timeout 300 tools/exp_scatter_calib > $R/timing.txt 2>&1; cat $R/timing.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf $ROOT/$R/p1 $ROOT/$R/p2
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$R/p1 -o p1 -- $ROOT/tools/exp_scatter_calib > $ROOT/$R/p1.log 2>&1; echo "rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_MISS_sum --output-format csv -d $ROOT/$R/p2 -o p2 -- $ROOT/tools/exp_scatter_calib > $ROOT/$R/p2.log 2>&1; echo "rc=$?" )
OUTDIR=$R python - <<'PY'
import collections, csv, glob, os
R = os.environ["OUTDIR"]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("p1", "p2"):
    for f in glob.glob(f"{R}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
N = 1 << 28
with open(f"{R}/counters.txt", "w") as o:
    o.write("kernel | FETCH_SIZE raw (KiB -> bytes) per access | TCC_EA0_RDREQ per access | of which 32-byte | TCC_REQ per access | TCC_MISS per access   (k_stream: per 128 bytes read)\n")
    for k in sorted(rows):
        v = {n: sorted(x)[len(x) // 2] for n, x in rows[k].items()}
        per = (8 << 30) / 128 if "k_stream" in k else N
        o.write("%-60s | %8.1f | %6.3f | %6.3f | %6.3f | %6.3f\n" % (k[:60], v.get("FETCH_SIZE", 0) * 1024 / per, v.get("TCC_EA0_RDREQ_sum", 0) / per,
                v.get("TCC_EA0_RDREQ_32B_sum", 0) / per, v.get("TCC_REQ_sum", 0) / per, v.get("TCC_MISS_sum", 0) / per))
print(open(f"{R}/counters.txt").read())
PY

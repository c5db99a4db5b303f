# the five rocprofv3 kernel-trace passes of tools/gpu/evidence.sh, alone (bash tools/gpu/r04_rocprof_only.sh r04)
RD=${1:-r04}
R=gpurun_out/$RD
mkdir -p $R
ROOT=$(pwd)
( cd /tmp && export TMPDIR=/tmp
  for spec in "u32_w7_unpack:prof_trace" "u32_mixed_unpack:prof_trace_mixed" "u64_w17_unpack:prof_trace_u64_unpack" "u64_w17_pack:prof_trace_u64_pack" "u32_w12_undelta_pack:prof_trace_undelta_pack"; do
    wl=${spec%%:*}; d=${spec##*:}
    rm -rf $ROOT/$R/$d
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/$d -o bench -- python $ROOT/bench.py --workload $wl --steps 10 --no-cpu-baseline --no-pmc --no-config5 > $ROOT/$R/$d.log 2>&1; echo "rocprof $wl rc=$?"
  done )

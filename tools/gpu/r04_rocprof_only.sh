# the five rocprofv3 kernel-trace passes of tools/gpu/evidence.sh, alone (bash tools/gpu/r04_rocprof_only.sh r04)
RD=${1:-r04}
R=gpurun_out/$RD
mkdir -p $R
ROOT=$(pwd)
( cd /tmp && export TMPDIR=/tmp
  for spec in "u32_w7_unpack:prof_trace" "u32_mixed_unpack:prof_trace_mixed" "u64_w17_unpack:prof_trace_u64_unpack" "u64_w17_pack:prof_trace_u64_pack" "u32_w12_undelta_pack:prof_trace_undelta_pack"; do
    wl=${spec%%:*}; d=${spec##*:}
    rm -rf $ROOT/$R/$d
    # which layout does --placement auto keep for this workload on this box?  Ask an unprofiled run, then profile with that layout passed
    # explicitly: the trace then holds nothing but the warm-ups and the timed launches of the kernel (no probe launches on the other layout)
    lay=$(timeout 300 python $ROOT/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-config5 --verify sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['config']['placement']; print('zoned' if 'LAYOUT_ZONED' in p else 'interleaved' if 'LAYOUT_INTERLEAVED' in p else 'separate')")
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/$d -o bench -- python $ROOT/bench.py --workload $wl --steps 10 --no-cpu-baseline --no-pmc --no-config5 --no-dispatch-check --placement ${lay:-separate} > $ROOT/$R/$d.log 2>&1; echo "rocprof $wl (placement ${lay:-separate}) rc=$?"
  done )

# Round-2 evidence run (on the GPU box, from the repo root): bench lines for every workload, rocprofv3 kernel trace
# of the headline bench, every-kernel sweeps, host-tier latency.  Outputs under gpurun_out/r02/ (scratch);
# tools/collect_profiles_r02.py copies the judged summaries into profiles/.
R=gpurun_out/r02
mkdir -p $R
ROOT=$(pwd)
timeout 900 python bench.py > $R/bench_u32w7.json 2> $R/bench.err; echo "bench rc=$?"
rm -f $R/bench_other.jsonl
for wl in u32_mixed_unpack u64_w17_unpack u64_w17_pack u32_w12_undelta_pack u32_w7_pack u16_w3_unpack; do
  timeout 400 python bench.py --workload $wl --steps 10 --cpu-seconds 3 >> $R/bench_other.jsonl 2>> $R/bench_other.err
done
( cd /tmp && export TMPDIR=/tmp && rm -rf $ROOT/$R/prof_trace $ROOT/$R/prof_trace_mixed && \
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/prof_trace -o bench -- python $ROOT/bench.py --steps 10 --no-cpu-baseline --no-pmc > $ROOT/$R/bench_under_rocprof.log 2>&1; echo "rocprof rc=$?"; \
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/prof_trace_mixed -o bench -- python $ROOT/bench.py --workload u32_mixed_unpack --steps 10 --no-cpu-baseline --no-pmc > $ROOT/$R/bench_mixed_under_rocprof.log 2>&1; echo "rocprof mixed rc=$?" )
for c in quick fused consume; do timeout 600 python tools/sweep.py --cases $c 2>&1 | grep -v amdgpu.ids > $R/sweep_$c.txt; done
timeout 120 tools/host_latency > $R/host_latency.txt 2>&1
find $R -name "*.csv" | head -20
cat $R/bench_u32w7.json

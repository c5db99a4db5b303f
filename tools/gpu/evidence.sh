# Evidence run of a round (on the GPU box, from the repo root):  bash tools/gpu/evidence.sh r03
# bench lines for every workload, rocprofv3 kernel trace of the bench, every-kernel sweeps, host-tier latency, the N-rank
# bench on one device (gloo and RCCL-attempt), the torch-free multi-device driver.  Outputs under gpurun_out/<round>/
# (scratch); tools/collect_profiles.py <round> copies the judged summaries into profiles/.
RD=${1:-r03}
R=gpurun_out/$RD
mkdir -p $R
ROOT=$(pwd)
timeout 900 python bench.py > $R/bench_u32w7.json 2> $R/bench.err; echo "bench rc=$?"
rm -f $R/bench_other.jsonl
for wl in u32_mixed_unpack u64_w17_unpack u64_w17_pack u32_w12_undelta_pack u32_w7_pack u16_w3_unpack; do
  timeout 400 python bench.py --workload $wl --steps 10 --cpu-seconds 3 >> $R/bench_other.jsonl 2>> $R/bench_other.err
done
# the headline is profiled WITHOUT the config-5 leg: both legs run the same kernel template (k_unpack_widths<u32>), and the
# stats average of a kernel must be comparable with the live average of ONE workload
( cd /tmp && export TMPDIR=/tmp && rm -rf $ROOT/$R/prof_trace $ROOT/$R/prof_trace_mixed && \
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/prof_trace -o bench -- python $ROOT/bench.py --steps 10 --no-cpu-baseline --no-pmc --no-config5 > $ROOT/$R/bench_under_rocprof.log 2>&1; echo "rocprof rc=$?"; \
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/prof_trace_mixed -o bench -- python $ROOT/bench.py --workload u32_mixed_unpack --steps 10 --no-cpu-baseline --no-pmc > $ROOT/$R/bench_mixed_under_rocprof.log 2>&1; echo "rocprof mixed rc=$?" )
for c in quick fused consume refbench batch; do timeout 600 python tools/sweep.py --cases $c 2>&1 | grep -v amdgpu.ids > $R/sweep_$c.txt; done
timeout 120 tools/host_latency > $R/host_latency.txt 2>&1
# N > 1 on this one device: 2 ranks, gloo control plane; then the same with the RCCL attempt (RCCL refuses two ranks on one
# GPU: the fallback path on real hardware)
timeout 600 python bench.py --gpus 2 --single-device --backend gloo --blocks 2000000 --steps 5 > $R/bench_2ranks_gloo.json 2> $R/bench_2ranks_gloo.err; echo "2 ranks gloo rc=$?"
timeout 900 python bench.py --gpus 2 --single-device --backend auto --nccl-probe-timeout 60 --blocks 2000000 --steps 5 > $R/bench_2ranks_auto.json 2> $R/bench_2ranks_auto.err; echo "2 ranks auto rc=$?"
# the driver's 8-rank shape on this one device: 8 processes, the 10 B-integer column split 1 220 704 + 7 x 1 220 703
timeout 900 python bench.py --gpus 8 --single-device --backend gloo --blocks 1000000 --steps 3 --warmup 1 > $R/bench_8ranks_one_device.json 2> $R/bench_8ranks_one_device.err; echo "8 ranks gloo rc=$?"
timeout 600 ./examples/multi_gpu_decode --steps 10 > $R/multi_gpu_decode.json 2> $R/multi_gpu_decode.err; echo "c driver rc=$?"
timeout 600 ./examples/multi_gpu_decode --steps 5 --replicas 2 --blocks 4000000 > $R/multi_gpu_decode_2threads.json 2>> $R/multi_gpu_decode.err; echo "c driver x2 rc=$?"
cat $R/bench_u32w7.json

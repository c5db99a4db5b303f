# Evidence run of a round (on the GPU box, from the repo root):  bash tools/gpu/evidence.sh r05
# bench lines for every workload (placement probed, every block verified), rocprofv3 kernel traces of BASELINE configs 2-5, every-kernel
# sweeps, host-tier latency, the N-rank bench on one device (gloo and RCCL-attempt), the torch-free multi-device driver.
# Outputs under gpurun_out/<round>/ (scratch); tools/collect_profiles.py <round> copies the judged summaries into profiles/.
RD=${1:-r06}
R=gpurun_out/$RD
mkdir -p $R
ROOT=$(pwd)
timeout 900 python bench.py > $R/bench_u32w7.json 2> $R/bench.err; echo "bench rc=$?"
rm -f $R/bench_other.jsonl
for wl in u32_mixed_unpack u32_mixed_random_unpack u64_w17_unpack u64_w17_pack u32_w12_undelta_pack u32_w7_pack u16_w3_unpack; do
  timeout 500 python bench.py --workload $wl --steps 10 --cpu-seconds 3 --no-config5 >> $R/bench_other.jsonl 2>> $R/bench_other.err; echo "$wl rc=$?"
done
# rocprofv3 kernel traces, one workload per pass (the stats average of a kernel must be comparable with the live average of ONE workload:
# configs 2 and 5 run the same kernel template, k_unpack_widths<u32>, hence --no-config5), with the layout --placement auto keeps on this box
( cd /tmp && export TMPDIR=/tmp
  for spec in "u32_w7_unpack:prof_trace" "u32_mixed_unpack:prof_trace_mixed" "u64_w17_unpack:prof_trace_u64_unpack" "u64_w17_pack:prof_trace_u64_pack" "u32_w12_undelta_pack:prof_trace_undelta_pack"; do
    wl=${spec%%:*}; d=${spec##*:}
    rm -rf $ROOT/$R/$d
    # which layout does --placement auto keep for this workload on this box?  Ask an unprofiled run, then profile with that layout passed
    # explicitly: the trace then holds nothing but the warm-ups and the timed launches of the kernel (no probe launches on the other layout)
    lay=$(timeout 300 python $ROOT/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-config5 --verify sample 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['config']['placement']; print('zoned' if 'LAYOUT_ZONED' in p else 'interleaved' if 'LAYOUT_INTERLEAVED' in p else 'separate')")
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/$d -o bench -- python $ROOT/bench.py --workload $wl --steps 10 --no-cpu-baseline --no-pmc --no-config5 --no-dispatch-check --placement ${lay:-separate} > $ROOT/$R/$d.log 2>&1; echo "rocprof $wl (placement ${lay:-separate}) rc=$?"
  done )
for c in quick fused consume refbench single; do timeout 900 python tools/sweep.py --cases $c 2>&1 | grep -v amdgpu.ids > $R/sweep_$c.txt; done
# mixed-width columns: each direction's buffers in a constructed layout (fl_column_pair_alloc: FL_LAYOUT_INTERLEAVED), then in plain tensors
timeout 900 python tools/sweep.py --cases mixed --placement interleaved 2>&1 | grep -v amdgpu.ids > $R/sweep_mixed.txt
timeout 900 python tools/sweep.py --cases mixed 2>&1 | grep -v amdgpu.ids > $R/sweep_mixed_separate.txt
timeout 900 python tools/ablayouts.py --cases headline,config5,config4,config3 2>&1 | grep -v amdgpu.ids > $R/ablayouts.txt
timeout 2400 python tools/sweep.py --cases allwidths --gb 8 --reps 5 2>&1 | grep -v amdgpu.ids > $R/sweep_allwidths.txt
# ... and with every row's buffers in a constructed pair (one process per element type: a pair's address ranges are never re-used)
bash tools/gpu/allwidths_constructed.sh $R/sweep_allwidths_constructed.txt > /dev/null
timeout 900 python tools/sweep.py --cases quick --placement interleaved --gb 12 2>&1 | grep -v amdgpu.ids > $R/sweep_quick_constructed.txt
timeout 600 python tools/sweep.py --cases batch --batch-all 2>&1 | grep -v amdgpu.ids > $R/sweep_batch.txt
timeout 120 tools/host_latency > $R/host_latency.txt 2>&1
timeout 600 python tools/pmc_single.py > $R/pmc_unpack_single.txt 2> $R/pmc_single.err; echo "pmc single rc=$?"
bash tools/gpu/sq_counters.sh tools/pmc_probe_mixed_delta.py $R/sq_mixed > $R/sq_mixed.log 2>&1; echo "sq counters rc=$?"
( time timeout 900 python -m pytest tests/test_gpu_full_check.py -m gpu -q ) > $R/full_check.txt 2>&1; echo "full check rc=$?"
# ... and shown to FAIL on the known-bad build (make -C fastlanes_amd/csrc BADSCAN=1: a patched copy of the sources, tests/checker/make_badscan_sources.py)
if [ -f fastlanes_amd/libfastlanes_amd_badscan.so ]; then
  FL_LIB=$ROOT/fastlanes_amd/libfastlanes_amd_badscan.so timeout 600 python -m pytest tests/test_gpu_full_check.py -m gpu -q -k under_load 2>&1 | grep -E "^E +AssertionError|^FAILED|passed|failed" > $R/full_check_badscan.txt; echo "known-bad build: $(tail -n 1 $R/full_check_badscan.txt)"
fi
# N > 1 on this one device: 2 ranks, gloo control plane; then the same with the RCCL attempt (RCCL refuses two ranks on one
# GPU: the fallback path on real hardware)
timeout 600 python bench.py --gpus 2 --single-device --backend gloo --blocks 2000000 --steps 5 > $R/bench_2ranks_gloo.json 2> $R/bench_2ranks_gloo.err; echo "2 ranks gloo rc=$?"
timeout 900 python bench.py --gpus 2 --single-device --backend auto --nccl-probe-timeout 60 --blocks 2000000 --steps 5 > $R/bench_2ranks_auto.json 2> $R/bench_2ranks_auto.err; echo "2 ranks auto rc=$?"
# the driver's 8-rank shape on this one device: 8 processes, the 10 B-integer column split 1 220 704 + 7 x 1 220 703; every rank verifies
# every block of both of its slices (--verify full is the default)
timeout 900 python bench.py --gpus 8 --single-device --backend gloo --blocks 1000000 --steps 3 --warmup 1 > $R/bench_8ranks_one_device.json 2> $R/bench_8ranks_one_device.err; echo "8 ranks gloo rc=$?"
timeout 600 ./examples/multi_gpu_decode --steps 10 > $R/multi_gpu_decode.json 2> $R/multi_gpu_decode.err; echo "c driver rc=$?"
timeout 600 ./examples/multi_gpu_decode --steps 5 --replicas 2 --blocks 4000000 > $R/multi_gpu_decode_2threads.json 2>> $R/multi_gpu_decode.err; echo "c driver x2 rc=$?"
cat $R/bench_u32w7.json

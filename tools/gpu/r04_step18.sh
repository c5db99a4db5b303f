# round 4, eighteenth GPU call: after the chain bodies became shared device functions: the whole GPU suite, the known-bad build against the full
# check, config 4 and the fused sweep
R=gpurun_out/r04v
mkdir -p $R
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $R/gpu_suite.txt 2>&1; echo "gpu suite rc=$?"; tail -n 5 $R/gpu_suite.txt | head -n 2
( time FL_LIB=$(pwd)/fastlanes_amd/libfastlanes_amd_badscan.so timeout 900 python -m pytest tests/test_gpu_full_check.py -m gpu -q -k "under_load" ) > $R/full_check_badscan.txt 2>&1; echo "bad build rc=$? (nonzero expected)"
grep -E "passed|failed|^E +AssertionError" $R/full_check_badscan.txt | head -6
timeout 600 python tools/sweep.py --cases fused 2>&1 | grep -v amdgpu.ids > $R/sweep_fused.txt; cat $R/sweep_fused.txt
timeout 300 python bench.py --workload u32_w12_undelta_pack 2> $R/bench_c4.err | tee $R/bench_c4.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CONFIG4', d['value'], d['roofline']['frac'], d['roofline'].get('placement_probe_GBps'))"

# round 4, eleventh GPU call: the final form (several blocks per wavefront only in the MIXED-WIDTH chain kernels): GPU suite, mixed
# sweep, the uniform chain kernels back where they were (fused sweep, bench config 4), and FoR's pack side of u8 with two blocks in
# flight (now possible: the reference is subtracted in the LDS image) against the table's kernel
R=gpurun_out/r04l
mkdir -p $R
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $R/gpu_suite.txt 2>&1; echo "gpu suite rc=$?"; tail -n 6 $R/gpu_suite.txt
timeout 900 python tools/sweep.py --cases mixed 2>&1 | grep -v amdgpu.ids > $R/sweep_mixed.txt; cat $R/sweep_mixed.txt
timeout 600 python tools/sweep.py --cases fused 2>&1 | grep -v amdgpu.ids > $R/sweep_fused.txt; cat $R/sweep_fused.txt
FL_LIB=$PWD/fastlanes_amd/libfastlanes_amd_full.so timeout 600 python tools/abnarrow.py --for 2>&1 | grep -v amdgpu.ids > $R/abnarrow_for.txt; cat $R/abnarrow_for.txt
timeout 300 python bench.py --workload u32_w12_undelta_pack 2> $R/bench_c4.err | tee $R/bench_c4.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CONFIG4', d['value'], d['roofline']['frac'], d['roofline'].get('placement_probe_GBps'))"

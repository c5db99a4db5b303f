# round 4, ninth GPU call: FoR / Delta over mixed-width columns (fl_<ty>_unfor_pack_widths, ..): parity, the full check under load,
# the mixed sweep; the uniform chain kernels must not have moved (fused sweep + bench config 4)
R=gpurun_out/r04i
mkdir -p $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mixed_width_columns or fused_transpose or delta_transpose" ) > $R/parity_new.txt 2>&1; echo "parity rc=$?"; tail -n 15 $R/parity_new.txt
( time timeout 900 python -m pytest tests/test_gpu_full_check.py -m gpu -q -x ) > $R/full_check.txt 2>&1; echo "full check rc=$?"; tail -n 8 $R/full_check.txt
timeout 900 python tools/sweep.py --cases mixed 2>&1 | grep -v amdgpu.ids > $R/sweep_mixed.txt; cat $R/sweep_mixed.txt
timeout 600 python tools/sweep.py --cases fused 2>&1 | grep -v amdgpu.ids > $R/sweep_fused.txt; cat $R/sweep_fused.txt
timeout 300 python bench.py --workload u32_w12_undelta_pack 2> $R/bench_c4.err | tee $R/bench_c4.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CONFIG4', d['value'], d['roofline']['frac'], d['roofline'].get('placement_probe_GBps'))"

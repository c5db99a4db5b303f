# round 4, nineteenth GPU call: "lane chains" for the narrow types' fused Delta decode with several blocks per wavefront (mixed-width columns,
# the small-array batch): parity, the full check under load, the mixed and batch sweeps
R=gpurun_out/r04w
mkdir -p $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "batch or mixed_width_columns or tile_map" ) > $R/parity.txt 2>&1; echo "parity rc=$?"; tail -n 12 $R/parity.txt
( time timeout 900 python -m pytest tests/test_gpu_full_check.py -m gpu -q -x ) > $R/full_check.txt 2>&1; echo "full check rc=$?"; tail -n 4 $R/full_check.txt | head -n 1
timeout 900 python tools/sweep.py --cases mixed 2>&1 | grep -v amdgpu.ids > $R/sweep_mixed.txt; grep -E "u16|u8" $R/sweep_mixed.txt
timeout 600 python tools/sweep.py --cases batch --batch-all 2>&1 | grep -v amdgpu.ids > $R/sweep_batch.txt; grep undelta $R/sweep_batch.txt | cut -c1-300

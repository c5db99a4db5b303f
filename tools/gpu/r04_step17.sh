# round 4, seventeenth GPU call: Delta over many small arrays (fl_<ty>_undelta_pack_batch / _transpose_delta_pack_batch): parity, then the batch sweep
R=gpurun_out/r04u
mkdir -p $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "batch or fused_transpose or delta_transpose or mixed_width_columns or tile_map" ) > $R/parity.txt 2>&1; echo "parity rc=$?"; tail -n 12 $R/parity.txt
timeout 600 python tools/sweep.py --cases batch --batch-all 2>&1 | grep -v amdgpu.ids > $R/sweep_batch.txt; cat $R/sweep_batch.txt

# round 4, tenth GPU call: the chain kernels of the narrow types with several blocks in flight per wavefront: the whole GPU suite
# (policy 2 forces them everywhere), the full check, the mixed sweep again, and the narrow-type re-sweep behind the dispatch table
# (cell-column vs chain at 3..8 waves, same buffers; FULL library).  TAG = box label.
TAG=${1:-a}
R=gpurun_out/r04j
mkdir -p $R
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $R/gpu_suite_$TAG.txt 2>&1; echo "gpu suite rc=$?"; tail -n 6 $R/gpu_suite_$TAG.txt
timeout 900 python tools/sweep.py --cases mixed 2>&1 | grep -v amdgpu.ids > $R/sweep_mixed_$TAG.txt; cat $R/sweep_mixed_$TAG.txt
FL_LIB=$PWD/fastlanes_amd/libfastlanes_amd_full.so timeout 1500 python tools/abchain.py 3 --all --types u16,u8 --gb 45 2>&1 | grep -v amdgpu > $R/abchain_narrow_$TAG.txt
grep -c MISMATCH $R/abchain_narrow_$TAG.txt; cat $R/abchain_narrow_$TAG.txt

# round 4, eighth GPU call: where do the family sweeps stand with the windowed map; compare kernels' occupancy again; the known-bad scan build against the full check
R=gpurun_out/r04h
mkdir -p $R
for c in quick fused consume; do timeout 600 python tools/sweep.py --cases $c 2>&1 | grep -v amdgpu.ids > $R/sweep_$c.txt; done
for a in "16 3" "8 3"; do timeout 200 tools/abbench thin $a 40000000 7 >> $R/abthin.txt 2>&1; done
( time FL_LIB=$(pwd)/fastlanes_amd/libfastlanes_amd_badscan.so timeout 900 python -m pytest tests/test_gpu_full_check.py -m gpu -q -k "under_load" ) > $R/full_check_badscan.txt 2>&1; echo "badscan rc=$? (nonzero expected)"
grep -E "^(FAILED|ERROR)|passed|failed|^E +AssertionError" $R/full_check_badscan.txt | head -30
( time timeout 900 python -m pytest tests/test_gpu_full_check.py -m gpu -q ) > $R/full_check.txt 2>&1; echo "full check rc=$?"; tail -n 5 $R/full_check.txt
cat $R/sweep_quick.txt $R/sweep_fused.txt $R/sweep_consume.txt $R/abthin.txt

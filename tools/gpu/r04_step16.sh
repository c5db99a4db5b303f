# round 4, sixteenth GPU call: the known-bad build (register scan + the deterministic sparse fault) against the full check and against the
# sampled under-load test of round 3; the library itself against both
R=gpurun_out/r04t
mkdir -p $R
( time FL_LIB=$(pwd)/fastlanes_amd/libfastlanes_amd_badscan.so timeout 900 python -m pytest tests/test_gpu_full_check.py -m gpu -q -k "under_load" ) > $R/full_check_badscan.txt 2>&1; echo "full check on the bad build rc=$? (nonzero expected)"
grep -E "^(FAILED|ERROR)|passed|failed|^E +AssertionError" $R/full_check_badscan.txt | head -20
( time FL_LIB=$(pwd)/fastlanes_amd/libfastlanes_amd_badscan.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "every_kernel_family_under_load or all_widths_vs_oracle or delta_transpose" ) > $R/parity_badscan.txt 2>&1; echo "sampled / small-size parity tests on the bad build rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $R/parity_badscan.txt | head -10
( time timeout 900 python -m pytest tests/test_gpu_full_check.py -m gpu -q ) > $R/full_check.txt 2>&1; echo "full check on the library rc=$?"; tail -n 4 $R/full_check.txt | head -n 1

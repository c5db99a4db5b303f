# round 4, fifth GPU call: the full-output check under load -- green on the library, and shown to FAIL on the known-bad r03 scan build
R=gpurun_out/r04e
mkdir -p $R
( time timeout 1500 python -m pytest tests/test_gpu_full_check.py -m gpu -q ) > $R/full_check.txt 2>&1; echo "full check rc=$?"; tail -6 $R/full_check.txt
( time FL_LIB=$(pwd)/fastlanes_amd/libfastlanes_amd_badscan.so timeout 900 python -m pytest tests/test_gpu_full_check.py -m gpu -q -k "under_load" ) > $R/full_check_badscan.txt 2>&1; echo "badscan rc=$? (nonzero expected)"
grep -E "^(FAILED|ERROR)|passed|failed|AssertionError: \(" $R/full_check_badscan.txt | head -40
FL_LIB=$(pwd)/fastlanes_amd/libfastlanes_amd_badscan.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "all_widths or delta_transpose or golden or under_load" > $R/parity_badscan.txt 2>&1; echo "sampled parity tests on the bad build rc=$?"; tail -3 $R/parity_badscan.txt

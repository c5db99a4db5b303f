# round 4, fifteenth GPU call: the final build three times through the whole GPU suite (no flakiness), and the known-bad scan build
# rebuilt from the refactored fl_chain.hpp against the full check (must still fail)
R=gpurun_out/r04s
mkdir -p $R
for i in 1 2 3; do ( time timeout 900 python -m pytest tests -m gpu -q -x ) > $R/gpu_suite_$i.txt 2>&1; echo "suite run $i rc=$?"; tail -n 4 $R/gpu_suite_$i.txt | head -n 1; done
( time FL_LIB=$(pwd)/fastlanes_amd/libfastlanes_amd_badscan.so timeout 900 python -m pytest tests/test_gpu_full_check.py -m gpu -q -k "under_load" ) > $R/full_check_badscan.txt 2>&1; echo "badscan rc=$? (nonzero expected)"
grep -E "^(FAILED|ERROR)|passed|failed|^E +AssertionError" $R/full_check_badscan.txt | head -20

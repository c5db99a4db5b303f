# The sweeps behind fastlanes_amd/csrc/fl_dispatch_table.inc.  Run this on TWO boxes (two gpurun calls), copy the two
# pairs of files to profiles/abfull_r03{a,b}.txt, profiles/abchain_r03{a,b}.txt and run tools/make_dispatch.py.
# Needs the FULL library (every cell-column instance):  make -C fastlanes_amd/csrc -j16 FULL=1
# Sizes: min(10 M blocks, 48 GB) per launch on torch allocations -- the size and allocation pattern of bench.py's workloads
# (which kernel / occupancy streams faster moves with both; profiles/abuniform_r03*.txt are the 16-GiB hipMalloc sweeps).
TAG=${1:-x}
R=gpurun_out/r03
mkdir -p $R
export FL_LIB=$PWD/fastlanes_amd/libfastlanes_amd_full.so
timeout 1500 python tools/abpack_full.py --all --gb 48 2>&1 | grep -v amdgpu > $R/abfull_$TAG.txt
timeout 1500 python tools/abchain.py 3 --all --gb 45 2>&1 | grep -v amdgpu > $R/abchain_big_$TAG.txt
grep -c MISMATCH $R/abfull_$TAG.txt $R/abchain_big_$TAG.txt
tail -n 2 $R/abfull_$TAG.txt; tail -n 2 $R/abchain_big_$TAG.txt

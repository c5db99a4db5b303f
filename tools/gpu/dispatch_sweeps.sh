# The sweeps behind fastlanes_amd/csrc/fl_dispatch_table.inc, for ONE box:  bash tools/gpu/dispatch_sweeps.sh <outdir>
# Run it on TWO boxes (two gpurun calls), copy the two pairs of files to profiles/abfull_<round>{a,b}.txt, profiles/abchain_<round>{a,b}.txt
# and run tools/make_dispatch.py on them.  Needs the FULL library (every cell-column instance):  make -C fastlanes_amd/csrc -j16 FULL=1
# Sizes: min(10 M blocks, 48 GB) per launch on torch allocations -- the size and allocation pattern of bench.py's workloads
# (which kernel / occupancy streams faster moves with both).  abchain.py --all covers Delta's kernels, the transposes, the fused
# transpose extensions and (since round 5) FoR's two bodies.
R=${1:-gpurun_out/dispatch}
mkdir -p $R
export FL_LIB=$PWD/fastlanes_amd/libfastlanes_amd_full.so
timeout 1500 python tools/abpack_full.py --all --gb 48 2>&1 | grep -v amdgpu > $R/abfull.txt
timeout 2400 python tools/abchain.py 3 --all --gb 45 2>&1 | grep -v amdgpu > $R/abchain.txt
grep -c MISMATCH $R/abfull.txt $R/abchain.txt
tail -n 2 $R/abfull.txt; tail -n 2 $R/abchain.txt

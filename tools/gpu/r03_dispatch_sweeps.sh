# The sweeps behind fastlanes_amd/csrc/fl_dispatch_table.inc.  Run this on TWO boxes (two gpurun calls), copy the two
# pairs of files to profiles/abuniform_r03{a,b}.txt, profiles/abchain_r03{a,b}.txt and run tools/make_dispatch.py.
# Needs the FULL library (every cell-column instance):  make -C fastlanes_amd/csrc -j16 FULL=1   and the tool linked to it:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fastlanes_amd/csrc -I include tools/abuniform.hip -L fastlanes_amd \
#         -lfastlanes_amd_full -Wl,-rpath,'$ORIGIN/../fastlanes_amd' -o tools/abuniform
TAG=${1:-x}
R=gpurun_out/r03
mkdir -p $R
timeout 900 tools/abuniform 5 > $R/abuniform_$TAG.txt 2>&1
FL_LIB=$PWD/fastlanes_amd/libfastlanes_amd_full.so timeout 1500 python tools/abchain.py 3 --all > $R/abchain_all_$TAG.txt 2>&1
grep -c MISMATCH $R/abuniform_$TAG.txt $R/abchain_all_$TAG.txt

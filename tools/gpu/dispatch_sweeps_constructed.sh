# The sweeps behind fl_dispatch_table.inc with every row's buffers in CONSTRUCTED pairs, for ONE box:
#     bash tools/gpu/dispatch_sweeps_constructed.sh <outdir> [types]
# (second half of round 6; tools/gpu/dispatch_sweeps.sh is the plain-memory form.)  One process per element type: a constructed pair's address
# ranges are never re-used within a process.  Needs the FULL library (make -C fastlanes_amd/csrc FULL=1).  24 GB per launch.
R=${1:-gpurun_out/dispatch_c}
TYPES=${2:-"u32 u64 u16 u8"}
mkdir -p $R
export FL_LIB=$PWD/fastlanes_amd/libfastlanes_amd_full.so
: > $R/abfull.txt; : > $R/abchain.txt
for ty in $TYPES; do
  timeout 1500 python tools/abpack_full.py --all --gb 24 --rounds 3 --constructed --types $ty 2>&1 | grep -v amdgpu >> $R/abfull.txt
  timeout 2400 python tools/abchain.py 3 --all --gb 24 --constructed --types $ty 2>&1 | grep -v amdgpu >> $R/abchain.txt
done
grep -c MISMATCH $R/abfull.txt $R/abchain.txt
grep -c "W=" $R/abfull.txt $R/abchain.txt
tail -n 2 $R/abfull.txt; tail -n 2 $R/abchain.txt

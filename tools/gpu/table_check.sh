# tools/gpu/table_check.sh <outfile>: tools/exp_table_check.py, one process per element type (needs the FULL library: the cell-column kernel of every (T, W))
O=${1:-gpurun_out/r06b/table_check.txt}
mkdir -p $(dirname $O); : > $O
export FL_LIB=$PWD/fastlanes_amd/libfastlanes_amd_full.so
for ty in u8 u16 u32 u64; do timeout 1500 python tools/exp_table_check.py --types $ty 2>&1 | grep -v amdgpu.ids >> $O; done
grep "^#" $O

# tools/gpu/allwidths_constructed.sh <outfile>: every (T, W) x 7 ops with each row's buffers in a constructed pair, one process per element type
# (a constructed pair's address ranges are never re-used within a process: fl_capi.hip reserve_fresh_range)
O=${1:-gpurun_out/r06b/sweep_allwidths_constructed.txt}
mkdir -p $(dirname $O); : > $O
for ty in u8 u16 u32 u64; do
  timeout 1500 python tools/sweep.py --cases allwidths --types $ty --gb 24 --reps 5 --placement interleaved 2>&1 | grep -v amdgpu.ids >> $O
done
grep "^# " $O | cut -c1-230

# tools/gpu/allwidths_constructed.sh <outfile>: every (T, W) x 7 ops with each row's buffers in a constructed pair (24 GB per row); one process
# per element type, the wide types in two width ranges (a constructed pair's address ranges are never re-used within a process:
# fl_capi.hip reserve_fresh_range -- on a box whose pools have to grow, 448 rows of 24 GB exhaust 96 TiB); one summary over all rows at the end
O=${1:-gpurun_out/r06c/sweep_allwidths_constructed.txt}
mkdir -p $(dirname $O); : > $O.rows
for spec in "u8 0 64" "u16 0 64" "u32 0 16" "u32 17 32" "u64 0 21" "u64 22 42" "u64 43 64"; do
  set -- $spec
  timeout 1500 python tools/sweep.py --cases allwidths --types $1 --wmin $2 --wmax $3 --gb 24 --reps 5 --placement interleaved 2>&1 | grep -v amdgpu.ids >> $O.rows
done
python tools/summarize_allwidths.py $O.rows > $O; rm -f $O.rows
grep "^# " $O | cut -c1-230

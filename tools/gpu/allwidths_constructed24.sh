# every (T, W) x 7 ops in constructed pairs at 24 GB per row (tools/gpu/allwidths_constructed.sh runs 12 GB)
O=${1:-gpurun_out/r06c/sweep_allwidths_constructed24.txt}
mkdir -p $(dirname $O); : > $O
for ty in u8 u16 u32 u64; do
  timeout 2000 python tools/sweep.py --cases allwidths --types $ty --gb 24 --reps 5 --placement interleaved 2>&1 | grep -v amdgpu.ids >> $O
done
grep "^# [a-z]" $O | cut -c1-110

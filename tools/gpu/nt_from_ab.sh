mkdir -p gpurun_out/r06b
for d in 4 8; do
for ty in u16 u32 u64; do
  FL_INTERNAL_NT_FROM_DIV=$d timeout 1500 python tools/sweep.py --cases allwidths --types $ty --gb 12 --reps 5 --placement interleaved 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06b/sweep_allwidths_constructed_nt$d.txt
done
done
echo done

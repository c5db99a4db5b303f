# Final sweep set -> gpurun_out/sweep_*.txt (copied into profiles/ by hand)
R=/root/repo/gpurun_out
mkdir -p $R
cd /root/repo
for c in quick fused orig consume widths; do
  timeout 900 python tools/sweep.py --cases $c --gb 16 --reps 5 2>&1 | grep -v amdgpu.ids > $R/sweep_$c.txt
done
python tools/sweep.py --cases host 2>&1 | grep -v amdgpu.ids > $R/sweep_host.txt
python tools/sweep.py --cases single 2>&1 | grep -v amdgpu.ids > $R/sweep_single.txt
python bench.py --workload u32_mixed_unpack --steps 10 > $R/bench_mixed.json 2>/dev/null
wc -l $R/sweep_*.txt

mkdir -p gpurun_out
R=/root/repo/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $R/pytest_gpu.log
timeout 600 python bench.py --workload u32_mixed_unpack --steps 10 > $R/bench_mixed.json 2>/dev/null; cut -c1-150 $R/bench_mixed.json

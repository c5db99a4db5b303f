mkdir -p gpurun_out
R=/root/repo/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $R/pytest_gpu.log
timeout 900 python tools/sweep.py --cases orig --json $R/sweep_orig.json 2>&1 | grep -v amdgpu.ids | tee $R/sweep_orig.txt

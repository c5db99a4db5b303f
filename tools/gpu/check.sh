mkdir -p gpurun_out
R=/root/repo/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $R/pytest_gpu.log
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HEADLINE', d['value'], d['roofline']['achieved'], d['roofline']['frac'])"
timeout 900 python tools/sweep.py --cases quick 2>&1 | grep -v amdgpu.ids | head -16

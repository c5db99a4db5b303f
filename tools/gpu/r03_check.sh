# round-3 quick check on the GPU box: GPU suite, default bench line, the 2-rank bench on one device (gloo and auto), the C driver
R=gpurun_out/r03
mkdir -p $R
timeout 1500 python -m pytest tests -m gpu -x -q > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $R/pytest_gpu.log
timeout 600 python bench.py > $R/bench_default.json 2> $R/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_default.json'))
print('HEADLINE', d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['correctness'], d['control_backend'])
c=d['config5_strong']; print('CONFIG5', c['value'], c['roofline_rank0']['frac'], c['correctness'])
PY
timeout 600 ./examples/multi_gpu_decode --steps 10 > $R/multi_gpu_decode_1gpu.json 2> $R/multi_gpu_decode_1gpu.err; echo "cdriver rc=$?"; cat $R/multi_gpu_decode_1gpu.json

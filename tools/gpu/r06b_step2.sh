# constructed-pair changes of round 6 (second session): three-class output rotation by position + whole-column map inside constructed pairs
R=gpurun_out/r06b; mkdir -p $R
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "interleaved or tile_map" ) > $R/step2_tests.txt 2>&1; echo "tests rc=$?"; tail -n 3 $R/step2_tests.txt
timeout 600 python tools/exp_pack_shape.py u32w7,u64w17,u32mixed,u64mixed 2>&1 | grep -v amdgpu.ids > $R/exp_pack_shape_after.txt; cat $R/exp_pack_shape_after.txt
timeout 600 python tools/sweep.py --cases mixed --placement interleaved 2>&1 | grep -v amdgpu.ids > $R/sweep_mixed_after.txt; cat $R/sweep_mixed_after.txt
for wl in u32_w7_pack u64_w17_pack; do timeout 600 python bench.py --workload $wl > $R/bench_$wl.json 2> $R/bench_$wl.err; echo "$wl rc=$?"; python - <<PY
import json
d=json.loads(open("$R/bench_$wl.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["config"]["workload"][:40], r["frac"], r.get("frac_of_bare_stream"), r.get("placement_probe_GBps"), d["config"]["placement"][-120:])
PY
done

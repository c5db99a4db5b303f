# tools/gpu/wg_ab.sh: the BASELINE workloads through builds with 256- (shipped), 128- and 64-thread workgroups (make WGSIZE=...), same box
R=gpurun_out/r06c; mkdir -p $R
for wg in ${WGS:-256 128 64}; do   # WGS="256 512": only workloads whose table entry is >= 6 waves launch at 512 (dynamic LDS <= 64 KiB)
  lib=$PWD/fastlanes_amd/libfastlanes_amd.so; [ $wg != 256 ] && lib=$PWD/fastlanes_amd/libfastlanes_amd_wg$wg.so
  FL_LIB=$lib timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "device_resident or all_widths_vs_oracle" 2>&1 | tail -1
  for wl in u32_w7_unpack u32_mixed_unpack u64_w17_pack; do
    FL_LIB=$lib timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-pmc --no-config5 --no-dispatch-check --verify sample > $R/wg${wg}_$wl.json 2> $R/wg${wg}_$wl.err
    python - <<PY
import json
try:
    d=json.loads(open("$R/wg${wg}_$wl.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("WG=$wg", "$wl", r["frac"], r.get("frac_of_bare_stream"), r.get("bare_stream_frac_of_peak"))
except Exception as e:
    print("WG=$wg $wl failed", e)
PY
  done
done

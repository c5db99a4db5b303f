# tools/gpu/ratio_layouts.sh: where should the OUTPUT's chunks come from as the read : write proportion moves?  (tools/exp_vmm, whole-column tile map)
mkdir -p gpurun_out/r06b
L=${EXP_L:-"A,1,BC,1;A,1,ABC,1;A,1,AB,1;A,1,B,1;A,1,BC,2;A,1,ABC,2"}
TAG=${EXP_TAG:-ratio}
for wl in "$@"; do
  set -- $wl
  EXP_LAYOUTS="$L" EXP_POLICY=1040187392 timeout 300 tools/exp_vmm 150 1024 $1 $2 2 > gpurun_out/r06b/vmm_${TAG}_$1_$2.txt 2>&1
  echo "== $1 $2"; tail -n $(( $(echo "$L" | tr -cd ';' | wc -c) + 4 )) gpurun_out/r06b/vmm_${TAG}_$1_$2.txt
done

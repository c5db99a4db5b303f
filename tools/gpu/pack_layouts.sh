# tools/gpu/pack_layouts.sh: pack u32 W=7 on layouts chosen by class (tools/exp_vmm, EXP_LAYOUTS), under the whole-column tile map and the windowed one
mkdir -p gpurun_out/r06b
L=${1:-"A,1,BC,2;A,1,B,1;A,1,BC,1;A,1,A,1;A,1,ABC,1;A,1,BC,-8;A,1,BC,4;AB,-2,C,1"}
N=${2:-8000000}
EXP_LAYOUTS="$L" EXP_POLICY=1040187392 timeout 500 tools/exp_vmm 150 1024 pack32w7 $N 2 > gpurun_out/r06b/vmm_pack_layouts_win31.txt 2>&1
tail -n 14 gpurun_out/r06b/vmm_pack_layouts_win31.txt
EXP_LAYOUTS="$L" timeout 500 tools/exp_vmm 150 1024 pack32w7 $N 2 > gpurun_out/r06b/vmm_pack_layouts_win16.txt 2>&1
tail -n 14 gpurun_out/r06b/vmm_pack_layouts_win16.txt

#!/bin/bash
# GPU box: the round-3 experiments around unpack_compare and the placement of thin streams, as one record of what was run (each
# step was its own gpurun call at the time; ~3 GPU-minutes in all).
#   bash tools/gpu/r03_compare_ab.sh [variant_lib.so ...]
# With library builds given (the build before a change, builds with an experiment macro: the macros named in the profiles'
# headers existed only for those runs) tools/ablibs.py times them against the current build on the same buffers.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03; mkdir -p $O
L=fastlanes_amd
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compare or consumer or under_load" > $O/cmp_tests.txt 2>&1; echo "tests rc=$?"; tail -n 3 $O/cmp_tests.txt
if [ $# -gt 0 ]; then
  CASES=u32:2,u32:4,u32:7,u32:10,u32:12,u32:16,u32:20,u32:24,u32:28,u32:31,u32:32,u64:4,u64:8,u64:12,u64:17,u64:24,u64:32,u64:40,u64:56,u64:64,u16:3,u16:9,u8:3
  timeout 400 python tools/ablibs.py 5 compare,sums $CASES $L/libfastlanes_amd.so "$@" 2>&1 | grep -v amdgpu.ids | tee $O/abcompare_libs.txt     # -> profiles/abcompare_butterfly_r03.txt, abcompare_maskstore_r03.txt
fi
for c in "u32 7" "u32 7 40000000" "u64 17" "u16 3 40000000" "u32 20"; do timeout 200 python tools/exp_zones_consumer.py $c; done 2>&1 | grep -v amdgpu.ids | tee $O/exp_zones_consumer.txt
for c in "compare u32 20" "compare u32 7" "undelta_pack u32 12" "undelta_pack u16 9"; do timeout 200 python tools/exp_thin_stream.py $c; done 2>&1 | grep -v amdgpu.ids | tee $O/exp_thin_stream.txt
timeout 300 python tools/exp_slab_size.py 2>&1 | grep -v amdgpu.ids | tee $O/exp_slab_size.txt                                                    # the three -> profiles/exp_thin_stream_r03.txt
timeout 300 python tools/exp_region_map.py 240 2>&1 | grep -v amdgpu.ids | tee $O/exp_region_map.txt
timeout 300 python tools/exp_region_map2.py 240 2>&1 | grep -v amdgpu.ids | tee $O/exp_region_map2.txt                                            # the two -> profiles/exp_region_map_r03.txt
timeout 300 env FL_LIB=$PWD/$L/libfastlanes_amd_full.so python tools/abpack_full.py 2>&1 | grep -v amdgpu.ids | tee $O/abpack_zoned.txt          # (make FULL=1 first) -> profiles/abpack_zoned_r03.txt

# round 4, twelfth GPU call: FoR's u8 bodies with two blocks in flight against the table's kernel, second box
R=gpurun_out/r04m
mkdir -p $R
FL_LIB=$PWD/fastlanes_amd/libfastlanes_amd_full.so timeout 600 python tools/abnarrow.py --for 2>&1 | grep -v amdgpu.ids > $R/abnarrow_for.txt; cat $R/abnarrow_for.txt

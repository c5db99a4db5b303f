#!/bin/bash
# GPU box: Delta's running-sum scan over the lane groups by DPP + v_permlane16/32_swap (fl_chain.hpp: scan_lane_groups) against the
# build before (Hillis-Steele over ds_bpermute): parity tests of every kernel that carries it, then same-buffer A/B.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03; mkdir -p $O
L=fastlanes_amd
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "delta or chain or all_widths or golden or fused or config" > $O/scan_tests.txt 2>&1; echo "tests rc=$?"; tail -n 3 $O/scan_tests.txt
timeout 300 python tools/ablibs.py 7 undelta,undelta_pack,undelta_pack_untranspose u32:12,u16:9,u8:4,u64:20,u32:7,u64:17 $L/libfastlanes_amd_prev.so $L/libfastlanes_amd.so > $O/abscan.txt 2>&1; echo "ablibs rc=$?"
cat $O/abscan.txt

#!/usr/bin/env python3
"""Makes the KNOWN-BAD sources of `make -C fastlanes_amd/csrc BADSCAN=1` (-> fastlanes_amd/libfastlanes_amd_badscan.so): a copy of the
library's sources with two deliberate defects patched into fl_chain.hpp.  Test scaffolding: the product headers carry none of this.

    python tests/checker/make_badscan_sources.py <csrc dir> <output dir>

1. Round 3's register-only form of Delta's lane-group scan (DPP + v_permlane16/32_swap instead of ds_bpermute).  It passed every
   per-(T, W) parity test and was wrong on ~3 % of the blocks of a u64 undelta_pack, differently on every run, with all CUs busy
   (profiles/abscan_r03.txt), and was dropped.  Its misbehaviour depends on instruction scheduling -- after fl_chain.hpp was cut into
   stages the same sequence stopped failing -- hence
2. a deterministic SPARSE fault: one wrong bit in one of every 4 099 blocks of a u64 undelta chain, the first at block 4 098 (beyond the
   small-size parity tests).  Sampled checks miss it; tests/test_gpu_full_check.py must not (profiles/full_check_r04.txt: exactly the
   121 faulty blocks).
Never loaded by anything but `FL_LIB=.../libfastlanes_amd_badscan.so pytest tests/test_gpu_full_check.py`.  Every replacement below must
match the current source exactly once, or this script fails: a refactor of fl_chain.hpp has to carry the patch along."""
import glob
import os
import shutil
import sys

src, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
for f in glob.glob(os.path.join(src, "*.hpp")) + glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.inc")):
    shutil.copy(f, out)
inc = os.path.abspath(os.path.join(src, "..", "..", "include"))
for f in glob.glob(os.path.join(out, "*.h*")):          # the copies sit two directories deeper: the public headers by absolute path
    t = open(f).read()
    if '"../../include/' in t:
        open(f, "w").write(t.replace('"../../include/', '"' + inc + "/"))
path = os.path.join(out, "fl_chain.hpp")
text = open(path).read()

BAD_SCAN = r'''
template <typename T> __device__ __forceinline__ Cell<T> scan_lane_groups_r03(Cell<T> v, unsigned lane)
{
    const bool odd_row = lane & 16u, upper_half = lane & 32u;
    auto upper_group_to_both = [](const Cell<T>& x) {
        u32x4 w = __builtin_bit_cast(u32x4, x), r;
        for (int k = 0; k < 4; ++k) r[k] = (uint32_t)__builtin_amdgcn_update_dpp((int)w[k], (int)w[k], 0x108 /* row_shl:8 */, 0xF, 0xF, false);
        return r;
    };
    {
        const u32x4 w = __builtin_bit_cast(u32x4, v);
        u32x4 below;
        for (int k = 0; k < 4; ++k) below[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[k], 0x118 /* row_shr:8 */, 0xF, 0xF, true);
        v = v.add(__builtin_bit_cast(Cell<T>, below));
    }
    {
        const u32x4 t = upper_group_to_both(v);
        u32x4 below;
        for (int k = 0; k < 4; ++k) {
            const auto sw = __builtin_amdgcn_permlane16_swap(t[k], t[k], false, false);
            below[k] = odd_row ? (uint32_t)sw[0] : 0u;
        }
        v = v.add(__builtin_bit_cast(Cell<T>, below));
    }
    {
        const u32x4 t = upper_group_to_both(v);
        u32x4 below;
        for (int k = 0; k < 4; ++k) {
            const auto sw = __builtin_amdgcn_permlane16_swap(t[k], t[k], false, false);
            const auto hf = __builtin_amdgcn_permlane32_swap(sw[1], sw[1], false, false);
            below[k] = upper_half ? (uint32_t)hf[0] : 0u;
        }
        v = v.add(__builtin_bit_cast(Cell<T>, below));
    }
    return v;
}
'''

GOOD = """        Cell<T> incl = x[R - 1];
        static_for<3>([&](auto S) {
            constexpr unsigned d = 1u << decltype(S)::value;
            incl = incl.add(cell_from_group_below<T>(incl, lane, d));
        });
        const Cell<T> excl = cell_from_group_below<T>(incl, lane, 1);
"""
BAD = "        const Cell<T> excl = scan_lane_groups_r03<T>(x[R - 1], lane).sub(x[R - 1]);      // KNOWN-BAD\n"
ANCHOR_FUNC = "// n x n element tile: in[j] = cell of row j (n lanes), out[e] = cell of lane e (n rows)"
ANCHOR_FAULT = "    if constexpr (FENCE_BEFORE_IMAGE) wave_lds_fence();\n    chain_stage_image<T, SNK>(x, lds, lane);\n    wave_lds_fence();\n    chain_stage_out<T, SNK>(a, blk, w, packed_at, lds, lane);"
FAULT = """    if constexpr (BODY == CHAIN_UNDELTA && sizeof(T) == 8) {
        if (blk % 4099u == 4098u && lane == 13u) x[0].x[0] ^= 1u;   // the planted sparse fault
    }
"""
for needle in (GOOD, ANCHOR_FUNC, ANCHOR_FAULT):
    if text.count(needle) != 1:
        sys.exit(f"make_badscan_sources.py: fl_chain.hpp no longer holds exactly one copy of:\n{needle}")
text = text.replace(GOOD, BAD).replace(ANCHOR_FUNC, BAD_SCAN + ANCHOR_FUNC).replace(ANCHOR_FAULT, FAULT + ANCHOR_FAULT)
open(path, "w").write(text)
print("known-bad sources in", out)

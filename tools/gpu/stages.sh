# The GPU calls of a round as NAMED STAGES (on the GPU box, from the repo root):
#     bash tools/gpu/stages.sh <round> <stage> [<stage> ...]
# Every stage writes under gpurun_out/<round>/ (scratch); tools/collect_profiles.py <round> copies the judged summaries into profiles/.
# (tools/gpu/evidence.sh <round> is the round's closing evidence run; the per-call scripts of rounds 3-4 are in the history up to 2c6e520.)
RD=$1; shift
R=gpurun_out/$RD
mkdir -p $R
ROOT=$(pwd)
for stage in "$@"; do
  echo "=== stage $stage"
  case $stage in
  suite)        # the whole GPU suite
    ( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $R/suite.txt 2>&1; echo "suite rc=$?"; tail -n 5 $R/suite.txt ;;
  parity_chain) # the tests that cover Delta / the transposes / mixed-width columns / the batch / the tile map
    ( time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "delta or chain or mixed_width or batch or tile_map or transpose" ) > $R/parity_chain.txt 2>&1; echo "parity rc=$?"; tail -n 6 $R/parity_chain.txt ;;
  full_check)
    ( time timeout 900 python -m pytest tests/test_gpu_full_check.py -m gpu -q -x ) > $R/full_check.txt 2>&1; echo "full check rc=$?"; tail -n 4 $R/full_check.txt ;;
  ab_r03)       # the round-3 build (eb76dff, tools/ab/libfastlanes_amd_r03.so) against HEAD, same buffers, interleaved
                # (the old build is not tracked: git worktree add build/r03_src eb76dff && make -C build/r03_src/fastlanes_amd/csrc -j8 &&
                #  mkdir -p tools/ab && cp build/r03_src/fastlanes_amd/libfastlanes_amd.so tools/ab/libfastlanes_amd_r03.so && git worktree remove --force build/r03_src)
    L="tools/ab/libfastlanes_amd_r03.so fastlanes_amd/libfastlanes_amd.so"
    { echo "# columns: round 3 (eb76dff) | HEAD;  BASELINE sizes (10 M blocks; config 5: 9 765 625), same buffers, round-robin"
      FL_AB_BLOCKS=10000000 timeout 600 python tools/ablibs.py 15 unpack u32:7,u64:17 $L
      FL_AB_BLOCKS=10000000 timeout 600 python tools/ablibs.py 15 pack u64:17 $L
      FL_AB_BLOCKS=10000000 timeout 600 python tools/ablibs.py 15 undelta_pack u32:12 $L
      FL_AB_BLOCKS=9765625 timeout 600 python tools/ablibs.py 15 unpack_widths u32:0 $L
      echo "# the 'quick' sweep's rows, ~8 GB of traffic each"
      timeout 900 python tools/ablibs.py 11 unpack,pack u32:7,u64:17,u16:3,u8:3 $L
      timeout 900 python tools/ablibs.py 11 undelta_pack,undelta_pack_untranspose u32:12,u16:9,u64:20,u8:4 $L
      timeout 900 python tools/ablibs.py 11 undelta u32:0,u64:0,u16:0,u8:0 $L
    } 2>&1 | grep -v amdgpu.ids > $R/ab_r03_vs_head.txt; cat $R/ab_r03_vs_head.txt ;;
  sq_mixed)     # SQ counters of Delta over mixed-width u8 / u16 columns
    bash tools/gpu/sq_counters.sh tools/pmc_probe_mixed_delta.py $R/sq_mixed > $R/sq_mixed.log 2>&1; cat $R/sq_mixed/sq_derived.txt ;;
  sweeps)       # quick / fused / mixed through the automatic dispatch
    for c in quick fused mixed; do timeout 900 python tools/sweep.py --cases $c 2>&1 | grep -v amdgpu.ids > $R/sweep_$c.txt; done
    cat $R/sweep_mixed.txt ;;
  sweep_mixed)
    timeout 900 python tools/sweep.py --cases mixed 2>&1 | grep -v amdgpu.ids > $R/sweep_mixed.txt; cat $R/sweep_mixed.txt ;;
  sweep_fused)
    timeout 900 python tools/sweep.py --cases fused 2>&1 | grep -v amdgpu.ids > $R/sweep_fused.txt; cat $R/sweep_fused.txt ;;
  allwidths)    # every (T, W) x {pack, unpack, unfor_pack, undelta_pack} through the automatic dispatch, one slab
    timeout 2400 python tools/sweep.py --cases allwidths --gb 8 --reps 5 2>&1 | grep -v amdgpu.ids > $R/sweep_allwidths.txt; tail -n 30 $R/sweep_allwidths.txt ;;
  single)       # unpack_single (a7)
    timeout 900 python tools/sweep.py --cases single 2>&1 | grep -v amdgpu.ids > $R/sweep_single.txt; cat $R/sweep_single.txt ;;
  window_ab)    # the tile-map window per (op, T): whole-column map vs 2^16-block windows vs 8-GiB windows, same buffers
    timeout 1500 python tools/abwindow.py ${FL_WINDOW_CASES:+--cases $FL_WINDOW_CASES} 2>&1 | grep -v amdgpu.ids > $R/window_ab.txt; cat $R/window_ab.txt ;;
  window_matrix) # every row of fl_window_table.inc x every type: the input of tools/make_window_table.py (one file per box)
    timeout 1500 python tools/abwindow.py --cases matrix --windows 31,16 --gb ${FL_MATRIX_GB:-48} 2>&1 | grep -v amdgpu.ids > $R/window_matrix.txt; cat $R/window_matrix.txt ;;
  dispatch_sweeps) # one box's input of tools/make_dispatch.py (needs libfastlanes_amd_full.so)
    bash tools/gpu/dispatch_sweeps.sh $R ;;
  chain_resume)  # the rest of a chain sweep that died: FL_CHAIN_START=u64:64 (round 5: the allocator fragmented at u64 W=64 on both boxes)
    FL_LIB=$PWD/fastlanes_amd/libfastlanes_amd_full.so timeout 2400 python tools/abchain.py 3 --all --gb 45 --start-at ${FL_CHAIN_START:-u64:64} 2>&1 | grep -v amdgpu > $R/abchain_resume.txt
    grep -c MISMATCH $R/abchain_resume.txt; tail -n 3 $R/abchain_resume.txt ;;
  chain_two_blocks) # the one- and two-blocks-per-wavefront forms of the wide types' undelta_pack, every width (input of make_dispatch.py; needs libfastlanes_amd_full.so)
    FL_LIB=$PWD/fastlanes_amd/libfastlanes_amd_full.so timeout 2400 python tools/abchain.py 3 --all --gb 45 --types u32,u64 --only undelta_pack,undelta_pack_2b 2>&1 | grep -v amdgpu > $R/abchain_two_blocks.txt
    grep -c MISMATCH $R/abchain_two_blocks.txt; tail -n 4 $R/abchain_two_blocks.txt ;;
  bench)
    timeout 900 python bench.py > $R/bench_u32w7.json 2> $R/bench.err; echo "bench rc=$?"; cat $R/bench_u32w7.json ;;
  *) echo "unknown stage $stage" ;;
  esac
done

# round 4, second GPU call: compare kernels vs bare thin-write streams + SQ counters; batch launch shapes for every type
R=gpurun_out/r04b
mkdir -p $R
timeout 600 python -m pytest tests -m gpu -x -q -k "batch or compare or cabi or c_abi or smoke" > $R/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $R/pytest.txt
for a in "16 3" "8 3" "32 7"; do timeout 200 tools/abbench thin $a 40000000 7 >> $R/abthin.txt 2>&1; echo "thin $a rc=$?"; done
P="$((2+256*8)),$((2+256*6)),$((2+256*5)),$((2+256*8+65536*2+16777216)),$((2+256*5+65536*2+16777216)),$((2+256*4+65536*2+16777216)),$((2+256*8+65536*4+16777216))"
timeout 900 python tools/sweep.py --cases batch --batch-all --batch-policies $P 2>&1 | grep -v amdgpu.ids > $R/sweep_batch_all.txt; echo "batch rc=$?"
FL_GB=6 bash tools/gpu/sq_counters.sh tools/pmc_probe_r03.py $R > $R/sq.log 2>&1; echo "sq rc=$?"
cat $R/abthin.txt $R/sweep_batch_all.txt $R/sq_derived.txt

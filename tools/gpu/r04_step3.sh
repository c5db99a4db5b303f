# round 4, third GPU call: windowed tile map for read-dominated streams; compare kernels at lower occupancies; batch shapes with interleaved timing
R=gpurun_out/r04c
mkdir -p $R
timeout 300 tools/abbench pack64 10000000 5 > $R/abpack_u64_windowed.txt 2>&1; echo "pack64 rc=$?"
for a in "16 3" "8 3" "32 7"; do timeout 200 tools/abbench thin $a 40000000 7 >> $R/abthin.txt 2>&1; echo "thin $a rc=$?"; done
P="$((2+256*8)),$((2+256*6)),$((2+256*5+65536*2+16777216)),$((2+256*8+65536*2+16777216)),$((2+256*8+65536*4+16777216))"
timeout 900 python tools/sweep.py --cases batch --batch-all --batch-policies $P 2>&1 | grep -v amdgpu.ids > $R/sweep_batch_all.txt; echo "batch rc=$?"
timeout 300 python -m pytest tests -m gpu -x -q -k "batch or compare" > $R/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $R/pytest.txt
cat $R/abpack_u64_windowed.txt $R/abthin.txt $R/sweep_batch_all.txt

# round 4, first GPU call: parity of the rewritten compare / batch kernels, config 3's pack leg vs bare streams, consumer + batch sweeps
R=gpurun_out/r04a
mkdir -p $R
timeout 900 python -m pytest tests -m gpu -x -q > $R/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $R/pytest.txt
timeout 300 tools/abbench pack64 10000000 5 > $R/abpack_u64.txt 2>&1; echo "abbench rc=$?"
timeout 600 python tools/sweep.py --cases consume 2>&1 | grep -v amdgpu.ids > $R/sweep_consume.txt; echo "consume rc=$?"
P="$((2+256*8)),$((2+256*5)),$((2+256*4)),$((2+256*6+65536*2)),$((2+256*5+65536*2+16777216)),$((2+256*4+65536*2+16777216)),$((2+256*3+65536*4+16777216))"
timeout 600 python tools/sweep.py --cases batch --batch-policies $P 2>&1 | grep -v amdgpu.ids > $R/sweep_batch.txt; echo "batch rc=$?"
cat $R/abpack_u64.txt $R/sweep_consume.txt $R/sweep_batch.txt

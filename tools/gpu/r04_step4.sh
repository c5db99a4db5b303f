# round 4, fourth GPU call: the windowed tile map against the whole-column map, every kernel family; parity of everything
R=gpurun_out/r04d
mkdir -p $R
timeout 900 python -m pytest tests -m gpu -x -q > $R/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $R/pytest.txt
timeout 1500 python tools/abwindow.py --reps 7 2>&1 | grep -v amdgpu.ids > $R/abwindow.txt; echo "abwindow rc=$?"
cat $R/abwindow.txt

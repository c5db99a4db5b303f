# round 4, thirteenth GPU call: the whole GPU suite with FL_CHECK_DEVICE=1 (the device checks must not refuse anything the suite does
# legitimately), then without it
R=gpurun_out/r04o
mkdir -p $R
( time FL_CHECK_DEVICE=1 timeout 1200 python -m pytest tests -m gpu -q -x -k "not check_device" ) > $R/gpu_suite_check_device.txt 2>&1; echo "FL_CHECK_DEVICE=1 suite rc=$?"; tail -n 8 $R/gpu_suite_check_device.txt
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $R/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -n 5 $R/gpu_suite.txt

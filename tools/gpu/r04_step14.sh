# round 4, fourteenth GPU call: the host tier with the pinned completion word (fl_capi.hip: HostCtx::wait_zero_copy): latency, the host-tier tests
R=gpurun_out/r04q
mkdir -p $R
timeout 120 tools/host_latency > $R/host_latency.txt 2>&1; cat $R/host_latency.txt
( time timeout 900 python -m pytest tests -m gpu -q -x -k "host or readme or reference_ffor or edge_cases or cpp_trait or plain_c or threads or survey" ) > $R/host_tests.txt 2>&1; echo "host tests rc=$?"; tail -n 5 $R/host_tests.txt

# round 3: the whole GPU suite, as it is and with the gemm_split default switched on (DSDF_GEMM_SPLIT=1)
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_suite.log 2>&1; rc=$?; tail -3 gpurun_out/r3_suite.log
[ $rc -eq 0 ] || { echo "FAILED: pytest rc $rc"; exit 1; }
DSDF_GEMM_SPLIT=1 timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_suite_split.log 2>&1; rc=$?; tail -3 gpurun_out/r3_suite_split.log
[ $rc -eq 0 ] || { echo "FAILED: pytest rc $rc"; exit 1; }

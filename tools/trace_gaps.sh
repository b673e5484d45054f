# kernel timeline of the timed steps: per-kernel start/end -> durations and gaps (tools/trace_gaps.py)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rm -rf $R/gpurun_out/trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $R/gpurun_out/trace.log 2>&1
echo "rc=$?"; tail -1 $R/gpurun_out/trace.log | cut -c1-300
python3 $R/tools/trace_gaps.py $(find $R/gpurun_out/trace -name '*kernel_trace.csv' | head -1)

# rocprofv3 per-kernel stats of the default bench (and of any extra bench args): gpurun_out/<tag>_bench_kernel_stats.csv
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=${1:-r04}; shift
cd /tmp && rm -rf $R/gpurun_out/stats && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats -- python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-profile --no-pmc --no-extras "$@" > $R/gpurun_out/stats.log 2>&1
cp $(find $R/gpurun_out/stats -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${T}_bench_kernel_stats.csv
cut -d, -f1-4,6-7 $R/gpurun_out/${T}_bench_kernel_stats.csv | cut -c1-160 | head -12

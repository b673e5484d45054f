# refresh the judged artefacts: MFMA counters + PMC traffic (tools/pmc.py), rocprofv3 kernel stats of the same command, bench line
# usage (GPU box): bash tools/final_profiles.sh r02 [extra bench args]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=${1:-r02}; shift
cd $R && python3 tools/pmc.py $T "$@" > gpurun_out/${T}_pmc.log 2>&1; tail -22 gpurun_out/${T}_pmc.log
cd /tmp && rm -rf $R/gpurun_out/stats && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-pmc --no-extras "$@" > $R/gpurun_out/stats.log 2>&1
cp $(find $R/gpurun_out/stats -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${T}_bench_kernel_stats.csv
cd $R && python3 bench.py "$@" > gpurun_out/${T}_bench.json.log 2>gpurun_out/${T}_bench.err; tail -1 gpurun_out/${T}_bench.json.log | cut -c1-3000

# refresh the judged artefacts: bench line, rocprofv3 kernel stats of the same command, PMC traffic
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=${1:-r01_seg}
cd $R && bash tools/pmc_traffic.sh > gpurun_out/${T}_pmc.log 2>&1 && cp gpurun_out/pmc_traffic.json profiles/${T}_pmc_traffic.json && tail -12 gpurun_out/${T}_pmc.log
cd /tmp && rm -rf $R/gpurun_out/stats && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $R/gpurun_out/stats.log 2>&1
cp $(find $R/gpurun_out/stats -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${T}_bench_kernel_stats.csv
cd $R && python bench.py > gpurun_out/${T}_bench.json.log 2>gpurun_out/${T}_bench.err; tail -1 gpurun_out/${T}_bench.json.log | cut -c1-2500

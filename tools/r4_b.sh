# round 4: after a kernel change -- the whole GPU suite (per-test durations), the dW launch's stamps, the headline / gemm_split / DP bench lines,
# config 4 and the shipped small networks
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -q -m gpu --durations=15 > gpurun_out/r4_suite_b.log 2>&1; rc=$?
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r4_suite_b.log | tail -20; [ $rc -eq 0 ] || { echo "FAILED: suite rc $rc"; exit 1; }
timeout -k 10 300 bash tools/lab_dw_stamps.sh || { echo "FAILED: dw stamps"; exit 1; }
one() { python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-extras "$@" 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],4), 'ms/step', round(d['value']/1e6,2), 'M pts/s', {k: (round(v['avg_us'],1), v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})"; }
{ one; one --config f32split; DSDF_FORCE_DP_PATH=1 one; } > gpurun_out/r4_b_bench.log 2>&1
cat gpurun_out/r4_b_bench.log
for n in 6x128 4x64 4x32; do one --network $n --steps 40; done > gpurun_out/r4_small_nets.log 2>&1
cat gpurun_out/r4_small_nets.log
timeout -k 10 300 python3 tools/extra_configs.py > gpurun_out/r4_extra_configs.log 2>&1; head -5 gpurun_out/r4_extra_configs.log
DSDF_FROWS=64 timeout -k 10 300 python3 tools/extra_configs.py 2>&1 | head -2 > gpurun_out/r4_extra_configs_frows64.log; cat gpurun_out/r4_extra_configs_frows64.log

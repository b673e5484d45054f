# round 4, first GPU call: the new data-parallel tests (RCCL rehearsal, K buckets), the dW launch's stamps + ride A/B, the 1-GPU
# cost of K = 1/2/4/8 gradient buckets, then the whole GPU suite with per-test durations.   usage (GPU box): bash tools/r4_a.sh
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "rccl or two_rank" > gpurun_out/r4_dist.log 2>&1; rc=$?
tail -5 gpurun_out/r4_dist.log; [ $rc -eq 0 ] || { echo "FAILED: dist tests rc $rc"; exit 1; }
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "phased" > gpurun_out/r4_phased.log 2>&1; rc=$?
tail -3 gpurun_out/r4_phased.log; [ $rc -eq 0 ] || { echo "FAILED: phased tests rc $rc"; exit 1; }
timeout -k 10 300 bash tools/lab_dw_stamps.sh || { echo "FAILED: dw stamps"; exit 1; }
one() { python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), 'ms/step', {k: (round(v['avg_us'],1), v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})"; }
{ one "default:"; DSDF_NO_RIDE=1 one "DSDF_NO_RIDE=1:"; } > gpurun_out/r4_ride_ab.log 2>&1
cat gpurun_out/r4_ride_ab.log
for b in 1 2 4 8; do DSDF_FORCE_DP_PATH=1 DSDF_AR_BUCKETS=$b one "DSDF_FORCE_DP_PATH=1 DSDF_AR_BUCKETS=$b:"; done > gpurun_out/r4_dp_buckets_one_gpu.log 2>&1
cat gpurun_out/r4_dp_buckets_one_gpu.log
timeout -k 10 1000 python3 -m pytest tests -q -m gpu --durations=60 > gpurun_out/r4_suite_a.log 2>&1; rc=$?
tail -75 gpurun_out/r4_suite_a.log; [ $rc -eq 0 ] || { echo "FAILED: suite rc $rc"; exit 1; }

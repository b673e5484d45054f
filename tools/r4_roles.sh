# round 4: the long-segment weight-gradient role -- parity subset + the shipped small specs
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "above_65536 or wave_private" > gpurun_out/r4_roles_tests.log 2>&1; rc=$?
tail -n 4 gpurun_out/r4_roles_tests.log; [ $rc -eq 0 ] || { echo "FAILED: tests rc $rc"; exit 1; }
one() { python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-extras "$@" 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],4), 'ms/step', round(d['value']/1e6,2), 'M pts/s', {k: (round(v['avg_us'],1), v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})"; }
{ for n in 4x32 4x64 6x128; do one --network $n; done; one --code-length 2 --scenes-per-batch 10 --samples 16000 --steps 30; one; } > gpurun_out/r4_roles_bench.log 2>&1
cat gpurun_out/r4_roles_bench.log

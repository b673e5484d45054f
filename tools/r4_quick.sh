# round 4: quick bench lines (headline, gemm_split, the shipped small nets)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
one() { python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-extras "$@" 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],4), 'ms/step', round(d['value']/1e6,2), 'M pts/s', {k: (round(v['avg_us'],1), v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})"; }
{ one; for n in 4x32 4x64 6x128; do one --network $n --steps 60; done; } > gpurun_out/r4_quick_bench.log 2>&1
cat gpurun_out/r4_quick_bench.log

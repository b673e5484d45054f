export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q -m gpu -k "golden or odd_shapes or fast_path or phased or two_rank or 65536 or properties or 512_scene or gemm_split" > gpurun_out/r4_d_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r4_d_tests.log; [ $rc -eq 0 ] || { grep -E "^E |Error" gpurun_out/r4_d_tests.log | head; exit 1; }
timeout -k 10 300 bash tools/lab_dw_stamps.sh || exit 1
one() { python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pmc --no-extras "$@" 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],4), 'ms/step', round(d['value']/1e6,2), 'M pts/s', {k: (round(v['avg_us'],1), v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})"; }
{ one; one --config f32split; one --config bf16split; } 2>&1 | tee gpurun_out/r4_d_bench.log
bash tools/r4_stats.sh r04s --config f32split

# round 4: the judged artefacts.  usage (GPU box): bash tools/r4_final.sh <part>   (parts keep a call under its time limit)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
one() { python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-extras "$@" 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],4), 'ms/step', round(d['value']/1e6,2), 'M pts/s', {k: (round(v['avg_us'],1), v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})"; }
case "$1" in
 a)  # headline fp32 + gemm_split: counters, rocprofv3 kernel stats, bench lines
  bash tools/final_profiles.sh r04 || exit 1
  bash tools/final_profiles.sh r04_f32split --config f32split || exit 1 ;;
 b)  # config 5 (+ split), the shipped shapes and small networks, the data-parallel call sequence with K buckets, config 4 & co
  for c in bf16 bf16split; do python3 bench.py --config $c > gpurun_out/r04_${c}_bench.json.log 2> gpurun_out/r04_${c}.err || exit 1; tail -1 gpurun_out/r04_${c}_bench.json.log | cut -c1-300; done
  for L in 2 16; do python3 bench.py --code-length $L --scenes-per-batch 10 --samples 16000 --steps 40 --warmup 5 > gpurun_out/r04_shipped_L$L.json.log 2> gpurun_out/r04_shipped_L$L.err || exit 1; tail -1 gpurun_out/r04_shipped_L$L.json.log | cut -c1-300; done
  for n in 6x128 4x64 4x32; do python3 bench.py --network $n --steps 40 --warmup 5 > gpurun_out/r04_net_$n.json.log 2> gpurun_out/r04_net_$n.err || exit 1; tail -1 gpurun_out/r04_net_$n.json.log | cut -c1-300; done
  for b in 1 2 4 8; do DSDF_FORCE_DP_PATH=1 DSDF_AR_BUCKETS=$b one; done > gpurun_out/r04_dp_buckets_one_gpu.log 2>&1; cat gpurun_out/r04_dp_buckets_one_gpu.log
  { python3 tools/extra_configs.py; bash tools/extra_numbers.sh; } > gpurun_out/r04_extra_configs.log 2>&1; tail -12 gpurun_out/r04_extra_configs.log ;;
 d)  # the data-parallel call sequence on one GPU with K gradient buckets (no collective)
  for b in 1 2 4 8; do DSDF_FORCE_DP_PATH=1 DSDF_AR_BUCKETS=$b one; done > gpurun_out/r04_dp_buckets_one_gpu.log 2>&1; cat gpurun_out/r04_dp_buckets_one_gpu.log ;;
 c)  # the whole GPU suite, as it is and with the gemm_split default switched on
  timeout -k 10 1100 python3 -m pytest tests -m gpu -q --durations=12 > gpurun_out/r04_gpu_suite.log 2>&1; rc=$?; tail -16 gpurun_out/r04_gpu_suite.log
  [ $rc -eq 0 ] || { echo "FAILED: pytest rc $rc"; exit 1; }
  DSDF_GEMM_SPLIT=1 timeout -k 10 1100 python3 -m pytest tests -m gpu -q > gpurun_out/r04_gpu_suite_gemm_split.log 2>&1; rc=$?; tail -3 gpurun_out/r04_gpu_suite_gemm_split.log
  [ $rc -eq 0 ] || { echo "FAILED: pytest (gemm_split) rc $rc"; exit 1; } ;;
esac

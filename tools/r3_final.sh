# round 3: every judged artefact from ONE box.  usage (GPU box): bash tools/r3_final.sh <part>   (parts keep a call under its time limit)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
case "$1" in
 a)  # headline fp32 + gemm_split: counters, rocprofv3 kernel stats, bench lines
  bash tools/final_profiles.sh r03 || exit 1
  bash tools/final_profiles.sh r03_f32split --config f32split || exit 1 ;;
 b)  # config 5 (+ split), the shipped shapes, the config-5 inference forward
  for c in bf16 bf16split; do python3 bench.py --config $c > gpurun_out/r03_${c}_bench.json.log 2> gpurun_out/r03_${c}.err || exit 1; tail -1 gpurun_out/r03_${c}_bench.json.log | cut -c1-400; done
  for L in 2 16; do python3 bench.py --code-length $L --scenes-per-batch 10 --samples 16000 --steps 40 --warmup 5 > gpurun_out/r03_shipped_L$L.json.log 2> gpurun_out/r03_shipped_L$L.err || exit 1; tail -1 gpurun_out/r03_shipped_L$L.json.log | cut -c1-300; done
  bash tools/bf16_decode_profiles.sh r03 || exit 1 ;;
 c)  # non-headline numbers quoted in DESIGN.md, fp32 MFMA and gemm_split; the data-parallel call sequence on one GPU
  { python3 tools/extra_configs.py; bash tools/extra_numbers.sh; } > gpurun_out/r03_extra_configs.log 2>&1 || exit 1
  { DSDF_GEMM_SPLIT=1 python3 tools/extra_configs.py; DSDF_GEMM_SPLIT=1 bash tools/extra_numbers.sh; } > gpurun_out/r03_extra_configs_gemm_split.log 2>&1 || exit 1
  for b in 1 2; do DSDF_FORCE_DP_PATH=1 DSDF_AR_BUCKETS=$b python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('DSDF_FORCE_DP_PATH=1 DSDF_AR_BUCKETS=$b:', round(d['ms_per_step'],4), 'ms/step', {k: (round(v['avg_us'],1), v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})"; done > gpurun_out/r03_dp_path_one_gpu.log 2>&1
  tail -4 gpurun_out/r03_extra_configs.log; tail -4 gpurun_out/r03_extra_configs_gemm_split.log; cat gpurun_out/r03_dp_path_one_gpu.log ;;
esac

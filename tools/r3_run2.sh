# round 3, GPU call 2: measured worst-element figures (-s), dW split-wait A/B, shipped shapes (L = 2 / 16, 10 x 16000)
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -s -k "full_size_step_vs_oracle or fast_path_train_step or shipped_experiment" > gpurun_out/r3_worst.log 2>&1 || exit 1
grep -E "worst|passed|failed" gpurun_out/r3_worst.log | tail -40
for round in 1 2; do bash tools/lab_split.sh; done > gpurun_out/r3_dw_ab.log 2>&1 || exit 1
cat gpurun_out/r3_dw_ab.log
for L in 2 16; do
  timeout -k 10 400 python bench.py --code-length $L --scenes-per-batch 10 --samples 16000 --steps 40 --warmup 5 > gpurun_out/r3_shipped_L$L.json.log 2> gpurun_out/r3_shipped_L$L.err || exit 1
  tail -c 600 gpurun_out/r3_shipped_L$L.json.log
done

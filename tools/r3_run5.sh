R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -k "segment_mode_odd or shipped_experiment or full_size_properties or fast_path" > gpurun_out/r3_t5.log 2>&1
rc=$?; tail -3 gpurun_out/r3_t5.log; [ $rc -le 1 ] || exit 1
for cfg in fp32 f32split; do python3 bench.py --config $cfg --scenes-per-batch 1 --samples 16384 --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('one scene x 16384, $cfg:', round(d['ms_per_step'],4), 'ms/step')"; done
for L in 2; do python3 bench.py --code-length $L --scenes-per-batch 10 --samples 16000 --steps 40 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shipped L=$L:', round(d['ms_per_step'],3), 'gemm_split', round(d['config']['gemm_split']['ms_per_step'],3))"; done

# round 3, GPU call 3: two-phase backward / two-bucket all-reduce, bf16x8 short-step pins, 1-GPU cost of the DP call sequence
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -s -k "two_phase or two_rank or bf16_8wave or bit_reproducible or decode_sdf_on_each or count_rule" > gpurun_out/r3_t3.log 2>&1
rc=$?; grep -E "worst|passed|failed|Error|error" gpurun_out/r3_t3.log | tail -40
[ $rc -le 1 ] || exit 1
for b in 1 2; do
  DSDF_FORCE_DP_PATH=1 DSDF_AR_BUCKETS=$b python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('force_dp buckets $b:', round(d['ms_per_step'],4), {k: (round(v['avg_us'],1), v['launches_per_step']) for k,v in d['roofline']['kernels'].items()})"
done
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fast path:', round(d['ms_per_step'],4))"

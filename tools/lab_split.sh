# lab: the step with NetworkSpecs gemm_split (bench.py --config f32split) against the fp32-MFMA build, per library variant
R=$GRAFT_REPO_ROOT; cd $R
for so in deepsdf_amd/libdsdf_hip.so $(ls tools/lab/variants/*.so 2>/dev/null); do
  for cfg in fp32 f32split; do
    DSDF_LIB_PATH=$R/$so python3 bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$so $cfg:', round(d['ms_per_step'],4), {k: round(v['avg_us'],1) for k,v in d['roofline']['kernels'].items()})"
  done
done

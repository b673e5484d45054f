# lab: bf16-forward variants (built on the CPU box into tools/lab/variants/)
R=$GRAFT_REPO_ROOT; cd $R
for so in deepsdf_amd/libdsdf_hip.so tools/lab/variants/*.so; do
  echo "== $so"
  DSDF_LIB_PATH=$R/$so python3 bench.py --config bf16 --steps 100 --warmup 10 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['roofline']['kernels']
print('  ms/step %.4f' % d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"
  DSDF_LIB_PATH=$R/$so python3 tools/lab_bf16_fwd.py 16384 2>/dev/null | head -1
done
DSDF_LIB_PATH=$R/deepsdf_amd/libdsdf_hip.so python3 tools/lab_bf16_fwd.py 16384 2>/dev/null
DSDF_LIB_PATH=$R/deepsdf_amd/libdsdf_hip.so LAB_BF16=1 python3 tools/lab_fused.py 16384 decode 2>/dev/null | tail -1

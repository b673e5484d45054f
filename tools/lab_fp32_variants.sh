R=$GRAFT_REPO_ROOT; cd $R
for so in deepsdf_amd/libdsdf_hip.so tools/lab/variants/*.so; do
  echo "== $so"
  for i in 1 2; do DSDF_LIB_PATH=$R/$so python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['roofline']['kernels']
print('  ms/step %.4f' % d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"; done
done

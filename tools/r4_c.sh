# round 4: narrow-net kernels -- parity on every kernel family, then the shipped small networks with and without them, stats of the headline
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_module_trainer.py -x -q -m gpu -k "golden or odd_shapes or fast_path or phased or 65536 or properties or eval or decode or trainer_end" > gpurun_out/r4_c_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4_c_tests.log; [ $rc -eq 0 ] || { echo "FAILED: tests rc $rc"; grep -E "^E |Error" gpurun_out/r4_c_tests.log | head -20; exit 1; }
one() { python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-extras "$@" 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', round(d['ms_per_step'],4), 'ms/step', round(d['value']/1e6,2), 'M pts/s', {k: (round(v['avg_us'],1), v['launches_per_step']) for k,v in r['kernels'].items()}, 'hbm frac', round(r.get('frac',0),3))"; }
for n in 6x128 4x64 4x32; do one --network $n; DSDF_NO_NARROW=1 one --network $n; done > gpurun_out/r4_small_nets.log 2>&1
cat gpurun_out/r4_small_nets.log
bash tools/r4_stats.sh r04a
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -n 1 | cut -c1-330

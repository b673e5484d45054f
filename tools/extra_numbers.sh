# non-headline numbers quoted in DESIGN.md: larger batches, end-to-end trainer, inference (decode) throughput
R=$GRAFT_REPO_ROOT; cd $R
for spb in 256 1024; do python bench.py --scenes-per-batch $spb --steps 20 --warmup 5 --no-cpu-baseline --no-profile | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('scenes/batch', $spb, 'pts/step', d['config']['points_per_step_per_gpu'], 'ms/step %.3f' % d['ms_per_step'], 'Mpts/s %.2f' % (d['value']/1e6))"; done
python tools/trainer_bench.py 2>&1 | tail -2
for n in 65536 262144 1048576; do python tools/lab_fused.py $n decode | tail -1; done

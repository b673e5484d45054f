for aux in 0 2 16; do
  echo "== aux $aux"; DSDF_LIB_PATH=$PWD/tools/lab/libdsdf_aux$aux.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: round(v['avg_us'],1) for k,v in d['roofline']['kernels'].items()})"
done

# lab: baseline stamps of the previous commit's forward in the training path (library built on the dev box)
R=$GRAFT_REPO_ROOT
DSDF_LIB_PATH=$R/tools/lab/libdsdf_head_lab.so DSDF_LAB_DBG=$R/gpurun_out/ff_dbg_head.bin python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile | cut -c1-60
python tools/lab_dbg.py $R/gpurun_out/ff_dbg_head.bin 256 | grep -E "k-loop median|epilogue median|total"

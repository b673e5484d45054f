# lab: per-layer stamps of the 8-wave bf16 forward (a -DDSDF_LAB build in tools/lab/variants/bf_lab.so): inference and training form
R=$GRAFT_REPO_ROOT
DSDF_LIB_PATH=$R/tools/lab/variants/bf_lab.so LAB_ONLY_BF16=1 DSDF_LAB_DBG=$R/gpurun_out/bf_dbg.bin python3 tools/lab_bf16_fwd.py 16384 | tail -n 1
python3 tools/lab_dbg8.py $R/gpurun_out/bf_dbg.bin 256
DSDF_LIB_PATH=$R/tools/lab/variants/bf_lab.so DSDF_LAB_DBG=$R/gpurun_out/bf_dbg_train.bin python3 bench.py --config bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extras > /dev/null 2>&1
python3 tools/lab_dbg8.py $R/gpurun_out/bf_dbg_train.bin 256

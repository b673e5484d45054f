# lab: per-layer stamps of the 8-wave bf16 forward (a -DDSDF_LAB build in tools/lab/variants/bf_lab.so): inference and training form
R=$GRAFT_REPO_ROOT
DSDF_LIB_PATH=$R/tools/lab/variants/bf_lab.so LAB_ONLY_BF16=1 DSDF_LAB_DBG=$R/gpurun_out/bf_dbg.bin python3 tools/lab_bf16_fwd.py 16384 | tail -n 1
python3 tools/lab_dbg8.py $R/gpurun_out/bf_dbg.bin 256

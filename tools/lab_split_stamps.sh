# lab: per-layer s_memtime stamps of the split-mode forward (k-loop / barrier / epilogue cycles) per -DDSDF_LAB library variant:
# the inference forward (decode) and the forward inside training steps
R=$GRAFT_REPO_ROOT; cd $R
for so in tools/lab/variants/lab_*.so; do
  echo "== $so  (inference forward)"
  DSDF_GEMM_SPLIT=1 DSDF_LIB_PATH=$R/$so DSDF_LAB_DBG=$R/gpurun_out/split_dbg.bin python3 tools/lab_fused.py 16384 decode > /dev/null 2>&1
  python3 tools/lab_dbg.py $R/gpurun_out/split_dbg.bin 256 | grep -E "k-loop median|epilogue median|total"
  echo "== $so  (training forward, segment mode)"
  DSDF_GEMM_SPLIT=1 DSDF_LIB_PATH=$R/$so DSDF_LAB_DBG=$R/gpurun_out/split_dbg_t.bin python3 tools/lab_train_stamps.py > /dev/null 2>&1
  python3 tools/lab_dbg.py $R/gpurun_out/split_dbg_t.bin 256 | grep -E "k-loop median|barrier median|epilogue median|total"
done

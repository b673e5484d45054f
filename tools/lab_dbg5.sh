R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-A B C D E}; do
  echo "== variant $v"
  DSDF_LIB_PATH=$R/tools/lab/libdsdf_lab.so DSDF_LAB_DBG=$R/gpurun_out/ff_dbg_v.bin python tools/lab_ff_after.py $v
  python tools/lab_dbg.py $R/gpurun_out/ff_dbg_v.bin 256 | grep -E "prologue|k-loop median|epilogue median"
done

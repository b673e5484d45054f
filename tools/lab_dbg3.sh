# lab: per-layer stamps of the fused forward (module_train) with N MB of freshly written (dirty) lines in front of it
R=$GRAFT_REPO_ROOT
for mb in 0 16 64 256; do
  echo "== dirty $mb MB"
  LAB_DIRTY_MB=$mb DSDF_LIB_PATH=$R/tools/lab/libdsdf_lab.so DSDF_LAB_DBG=$R/gpurun_out/ff_dbg_m.bin python tools/lab_fused.py 16384 module_train | tail -1
  python tools/lab_dbg.py $R/gpurun_out/ff_dbg_m.bin 256 | grep -E "prologue|k-loop median|epilogue median"
done

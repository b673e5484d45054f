# lab: forward epilogue with stores but WITHOUT the dropout hash (module path, eval mode) vs with it
R=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDSDF_LAB -o $R/tools/lab/libdsdf_lab.so $R/deepsdf_amd/csrc/dsdf_api.hip 2>/dev/null
for mode in module_eval module_train; do
  DSDF_LIB_PATH=$R/tools/lab/libdsdf_lab.so DSDF_LAB_DBG=$R/gpurun_out/ff_dbg_m.bin python tools/lab_fused.py 16384 $mode | tail -1
  python tools/lab_dbg.py $R/gpurun_out/ff_dbg_m.bin 256 | grep -E "k-loop median|epilogue median"
done

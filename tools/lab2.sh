python tools/lab_fused.py 16384 module_train | grep ablate
python tools/lab_fused.py 16384 decode | grep ablate
export DSDF_LIB_PATH=$PWD/tools/lab/libdsdf_lab.so
DSDF_LAB_DBG=$PWD/gpurun_out/dbg0.bin python tools/lab_fused.py 16384 module_train | grep ablate
python tools/lab_dbg.py gpurun_out/dbg0.bin 256

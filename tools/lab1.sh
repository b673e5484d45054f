for a in 0 1 2 3 5 7; do DSDF_LAB_ABLATE=$a python tools/lab_fused.py 16384 module_train; done
DSDF_LAB_ABLATE=0 python tools/lab_fused.py 8192 module_train
DSDF_LAB_ABLATE=0 python tools/lab_fused.py 16384 decode
DSDF_LAB_ABLATE=0 python tools/lab_fused.py 16384 module_eval
DSDF_LAB_ABLATE=0 python tools/lab_fused.py 32768 module_train

for a in 0 8 16 24 0; do DSDF_LAB_ABLATE=$a python tools/lab_fused.py 16384 decode | grep ablate; done
DSDF_LAB_ABLATE=0 python tools/lab_fused.py 32768 decode | grep ablate
DSDF_LAB_ABLATE=0 python tools/lab_fused.py 65536 decode | grep ablate

# lab: variants of the 8-wave bf16 forward (built on the CPU box into tools/lab/variants/): decode_latent, 16384 points
R=$GRAFT_REPO_ROOT; cd $R
for so in deepsdf_amd/libdsdf_hip.so tools/lab/variants/*.so; do
  echo "== $so"
  DSDF_LIB_PATH=$R/$so LAB_ONLY_BF16=1 python3 tools/lab_bf16_fwd.py 16384 2>/dev/null | tail -n 1
  DSDF_LIB_PATH=$R/$so LAB_ONLY_BF16=1 python3 tools/lab_bf16_fwd.py 1048576 2>/dev/null | tail -n 1
done

# lab: run-to-run differences / outliers + speed of the 8-wave bf16 forward per library variant
R=$GRAFT_REPO_ROOT; cd $R
for so in tools/lab/variants/*.so; do
  echo "== $so"
  DSDF_LIB_PATH=$R/$so timeout -k 10 200 python3 tools/lab_bf16x8_err.py 2>/dev/null | tail -n 5 | cut -c1-300
  DSDF_LIB_PATH=$R/$so LAB_ONLY_BF16=1 python3 tools/lab_bf16_fwd.py 16384 2>/dev/null | tail -n 1
  DSDF_LIB_PATH=$R/$so LAB_ONLY_BF16=1 python3 tools/lab_bf16_fwd.py 1048576 2>/dev/null | tail -n 1
done

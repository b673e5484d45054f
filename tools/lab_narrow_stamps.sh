# lab: s_memtime stamps of the narrow-net MERGED forward + backward launch inside real training steps of the shipped small specs
# (tools/lab/libdsdf_lab.so = a -DDSDF_LAB build)
R=$GRAFT_REPO_ROOT; cd $R
for n in 4x32 4x64 6x128; do
  nl=3; [ $n = 6x128 ] && nl=5
  echo "== $n"
  DSDF_LIB_PATH=$R/tools/lab/libdsdf_lab.so DSDF_LAB_MDBG=$R/gpurun_out/nn_mdbg.bin python3 bench.py --network $n --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-pmc --no-extras | cut -c1-60
  python3 tools/lab_mdbg.py $R/gpurun_out/nn_mdbg.bin 2500 $nl
done
echo "== 8x512"
DSDF_LIB_PATH=$R/tools/lab/libdsdf_lab.so DSDF_LAB_MDBG=$R/gpurun_out/nn_mdbg.bin python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-pmc --no-extras 2>/dev/null | cut -c1-60
python3 tools/lab_mdbg.py $R/gpurun_out/nn_mdbg.bin 256 8

# lab: s_memtime stamps of the fused forward inside a real training step (bench workload)
set -e
R=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDSDF_LAB -o $R/tools/lab/libdsdf_lab.so $R/deepsdf_amd/csrc/dsdf_api.hip
DSDF_LIB_PATH=$R/tools/lab/libdsdf_lab.so DSDF_LAB_DBG=$R/gpurun_out/ff_dbg.bin python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile | cut -c1-200
python tools/lab_dbg.py $R/gpurun_out/ff_dbg.bin 256

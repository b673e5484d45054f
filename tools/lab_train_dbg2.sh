# as lab_train_dbg.sh (library already built there) but 2 workgroups per CU: is the second one (warm I-cache) faster?
R=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDSDF_LAB -o $R/tools/lab/libdsdf_lab.so $R/deepsdf_amd/csrc/dsdf_api.hip 2>/dev/null
DSDF_LIB_PATH=$R/tools/lab/libdsdf_lab.so DSDF_LAB_DBG=$R/gpurun_out/ff_dbg.bin python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --scenes-per-batch 128 | cut -c1-100
python - <<'PY'
import numpy as np, os
d = np.fromfile(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/ff_dbg.bin", dtype=np.uint64).reshape(8192, 64)[:512].astype(np.int64)
order = np.argsort(d[:, 62])
first, second = order[:256], order[256:]
for name, idx in (("first 256 WGs to start", first), ("last 256 WGs to start", second)):
    x = d[idx]
    k = np.stack([x[:, 1 + 3*l] - (x[:, 0] if l == 0 else x[:, 3*l]) for l in range(7)], 1)
    e = np.stack([x[:, 3 + 3*l] - x[:, 2 + 3*l] for l in range(7)], 1)
    print(name, "prologue", np.median(x[:, 0] - x[:, 62]), "\n  k-loop", np.median(k, 0), "\n  epilogue", np.median(e, 0))
PY

# -DDSDF_LAB build: per-wave s_memrealtime stamps (100 MHz) of the dW launch inside real training steps -- when does the last MFMA
# item end, when does each role workgroup end, and how long do the three role kinds take on the spare workgroups?
# usage (GPU box): bash tools/lab_dw_stamps.sh [bench args]     -> gpurun_out/dw_stamps.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/tools/lab/variants $R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DDSDF_LAB -o $R/tools/lab/variants/lab.so $R/deepsdf_amd/csrc/dsdf_api.hip || exit 1
DSDF_LIB_PATH=$R/tools/lab/variants/lab.so DSDF_LAB_DWDBG=$R/gpurun_out/dw_dbg.bin python3 $R/bench.py --steps 6 --warmup 4 --init-steps 10 --no-cpu-baseline --no-pmc --no-extras "$@" > /dev/null || exit 1
python3 - <<'PY' | tee $R/gpurun_out/dw_stamps.log
import numpy as np, os
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
d = np.fromfile(R + "/gpurun_out/dw_dbg.bin", dtype=np.uint64).reshape(1024, 4, 8).astype(np.int64)
live = d[:, :, 0] > 0
t0 = d[:, :, 0][live].min()
role = (d[:, 0, 2] == 1) & live[:, 0]
item = ~role & live[:, 0]
us = lambda x: (x - t0) / 100.0
print(f"workgroups: {int(item.sum())} item, {int(role.sum())} role (stamps in us after the first wave's start; last launch of the run)")
iw = d[item]
starts, ends = us(iw[:, :, 0][iw[:, :, 0] > 0]), us(iw[:, :, 1][iw[:, :, 1] > 0])
print(f"item waves: start median {np.median(starts):.1f} max {starts.max():.1f}; end median {np.median(ends):.1f} p90 {np.percentile(ends, 90):.1f} max {ends.max():.1f}")
if role.any():
    rw = d[role][:, 0]
    e = us(rw[:, 1])
    print(f"role workgroups: start median {np.median(us(rw[:, 0])):.1f}; end min {e.min():.1f} median {np.median(e):.1f} max {e.max():.1f}")
    for k, name in enumerate(("head partials (reduce_rows)", "x0 columns of dW (seg_dw)", "per-segment latent gradient (seg_latgrad)")):
        print(f"  time inside {name}: median {np.median(rw[:, 3 + k]) / 100.0:.1f} us, max {rw[:, 3 + k].max() / 100.0:.1f} us per role workgroup")
PY

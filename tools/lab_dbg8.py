"""Lab: per-layer phase times of the 8-wave bf16 forward from its s_memtime stamps (DSDF_LAB build + DSDF_LAB_DBG=file).
usage: python tools/lab_dbg8.py file n_workgroups"""
import numpy as np, sys
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(8192, 64)[:int(sys.argv[2])].astype(np.int64)
A, B = d[:, :32], d[:, 32:]
np.set_printoptions(linewidth=220)
nl = 8
print("(s_memtime ticks: 100 MHz)")
print("A  begin+K (incl. mid barrier):", np.median(np.stack([A[:, 4*l+1] - A[:, 4*l] for l in range(nl)], 1), 0))
print("A  epilogue                   :", np.median(np.stack([A[:, 4*l+2] - A[:, 4*l+1] for l in range(nl)], 1), 0))
print("A  end barrier wait           :", np.median(np.stack([A[:, 4*l+3] - A[:, 4*l+2] for l in range(nl)], 1), 0))
print("B  epilogue (of l-1)          :", np.median(np.stack([B[:, 4*l+1] - B[:, 4*l] for l in range(nl)], 1), 0))
print("B  barrier wait               :", np.median(np.stack([B[:, 4*l+2] - B[:, 4*l+1] for l in range(nl)], 1), 0))
print("B  begin+K                    :", np.median(np.stack([B[:, 4*l+3] - B[:, 4*l+2] for l in range(nl)], 1), 0))
print("B  end barrier wait           :", np.median(np.stack([B[:, 4*(l+1)] - B[:, 4*l+3] for l in range(nl-1)], 1), 0))
print("layer period (A)              :", np.median(np.stack([A[:, 4*(l+1)] - A[:, 4*l] for l in range(nl-1)], 1), 0))
print("total A median/max:", np.median(A[:, 4*nl-1] - A[:, 0]), (A[:, 4*nl-1] - A[:, 0]).max())

"""Lab: phase times of the MERGED forward + backward launch from its s_memtime stamps (-DDSDF_LAB library + DSDF_LAB_MDBG=file).
usage: python tools/lab_mdbg.py file n_workgroups n_hidden_layers   (cycles of the 100 MHz counter x 24 = shader cycles at 2.4 GHz is NOT
applied: the numbers are raw counter ticks)"""
import sys
import numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(8192, 64)
nwg, nl = int(sys.argv[2]), int(sys.argv[3])
d = d[:nwg].astype(np.int64)
np.set_printoptions(linewidth=200)
med = lambda x: np.median(x, 0)
k = np.stack([d[:, 1 + 3 * l] - (d[:, 0] if l == 0 else d[:, 3 * l]) for l in range(nl)], 1)
b = np.stack([d[:, 2 + 3 * l] - d[:, 1 + 3 * l] for l in range(nl)], 1)
e = np.stack([d[:, 3 + 3 * l] - d[:, 2 + 3 * l] for l in range(nl)], 1)
print("forward  k-loop  :", med(k)); print("forward  barrier :", med(b)); print("forward  epilogue:", med(e))
print("forward  total   :", med(d[:, 3 * nl] - d[:, 0]))
print("fwd end -> bwd start:", med(d[:, 32] - d[:, 3 * nl]), " head:", med(d[:, 33] - d[:, 32]))
nb = nl - 1
kb = np.stack([d[:, 34 + 3 * i] - (d[:, 33] if i == 0 else d[:, 33 + 3 * i]) for i in range(nb)], 1)
bb = np.stack([d[:, 35 + 3 * i] - d[:, 34 + 3 * i] for i in range(nb)], 1)
eb = np.stack([d[:, 36 + 3 * i] - d[:, 35 + 3 * i] for i in range(nb)], 1)
print("backward k-loop  :", med(kb)); print("backward barrier :", med(bb)); print("backward epilogue:", med(eb))
tot = d[:, 33 + 3 * nb] - d[:, 0]
print("workgroup total median / max:", med(tot), tot.max(), " launch span:", d[:, 33 + 3 * nb].max() - d[:, 0].min())

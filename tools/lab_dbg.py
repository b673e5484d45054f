import numpy as np, sys
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(8192, 64)
nwg = int(sys.argv[2]); nl = int(sys.argv[3]) if len(sys.argv) > 3 else 8
d = d[:nwg].astype(np.int64)
t0 = d[:, 0]
print("start spread (cycles@100MHz?):", t0.max() - t0.min())
k = np.stack([d[:, 1 + 3*l] - (d[:, 0] if l == 0 else d[:, 3*l]) for l in range(nl)], 1)      # k-loop duration
b = np.stack([d[:, 2 + 3*l] - d[:, 1 + 3*l] for l in range(nl)], 1)                               # barrier wait
e = np.stack([d[:, 3 + 3*l] - d[:, 2 + 3*l] for l in range(nl)], 1)                               # epilogue
tot = d[:, 3*nl] - d[:, 0]
np.set_printoptions(linewidth=200)
print("prologue: start->rows %s  rows->slab ready %s" % (np.median(d[:, 61] - d[:, 62]), np.median(d[:, 0] - d[:, 61])))
print("k-loop median per layer :", np.median(k, 0))
print("k-loop max per layer    :", k.max(0))
print("barrier median per layer:", np.median(b, 0))
print("epilogue median per layer:", np.median(e, 0))
print("total median/max:", np.median(tot), tot.max())
for w in (0, 1, 256, 257, 511):
    if w < nwg: print("wg", w, "start", d[w,0]-t0.min(), "k", k[w], "e", e[w])

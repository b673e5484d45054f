"""Lab (CPU): accuracy of an fp32 GEMM emulated on bf16 matrix cores by operand splitting (x = hi + mid + lo, bf16 terms, products
accumulated in fp32) against the plain fp32 GEMM, both measured against fp64.  K = 512, the decoder's shapes."""
import numpy as np, torch
torch.manual_seed(0)
def bf(x): return x.to(torch.bfloat16).to(torch.float32)
def split3(x):
    h = bf(x); r = x - h; m = bf(r); l = bf(r - m)
    return h, m, l
M, K, N = 4096, 512, 512
a = torch.relu(torch.randn(M, K)); w = torch.randn(N, K) * 0.05
ref = (a.double() @ w.double().T)
def err(c): return float((c.double() - ref).norm() / ref.norm()), float((c.double() - ref).abs().max() / ref.abs().max())
print("fp32 GEMM                     rel %.2e  max %.2e" % err(a @ w.T))
ah, am, al = split3(a); wh, wm, wl = split3(w)
print("split exact? a: %.1e  w: %.1e" % (float((ah + am + al - a).abs().max()), float((wh + wm + wl - w).abs().max())))
def mm(x, y): return (x.double() @ y.double().T).float()      # each bf16 x bf16 product is exact in fp32; model the fp32 accumulation coarsely
terms = {"hh": mm(ah, wh), "hm": mm(ah, wm), "mh": mm(am, wh), "mm": mm(am, wm), "hl": mm(ah, wl), "lh": mm(al, wh),
         "ml": mm(am, wl), "lm": mm(al, wm), "ll": mm(al, wl)}
for name, keys in (("3 products (hh hm mh)", ["hh", "hm", "mh"]), ("6 products (+ mm hl lh)", ["hh", "hm", "mh", "mm", "hl", "lh"]),
                   ("9 products", list(terms))):
    c = sum(terms[k] for k in sorted(keys, key=lambda k: float(terms[k].abs().mean())))    # small terms first
    print("%-28s rel %.2e  max %.2e" % ((name,) + err(c)))
# fp32 accumulation inside the k-loop: chunked fp32 sums of 16-deep partial products, as the MFMA would do
def chunked(xs, ys):
    c = torch.zeros(M, N)
    for k0 in range(0, K, 16):
        for x, y in zip(xs, ys):
            c += (x[:, k0:k0 + 16].double() @ y[:, k0:k0 + 16].double().T).float()
    return c
c6 = chunked([al, ah, am, am, ah, ah], [wh, wl, wm, wh, wm, wh])
print("%-28s rel %.2e  max %.2e" % (("6 products, fp32 accumulate per 16-deep step",) + err(c6)))
cf = torch.zeros(M, N)
for k0 in range(0, K, 2):
    cf += (a[:, k0:k0 + 2].double() @ w[:, k0:k0 + 2].double().T).float()
print("%-28s rel %.2e  max %.2e" % (("fp32 MFMA model (accumulate per 2-deep step)",) + err(cf)))

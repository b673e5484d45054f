"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: duration of each kernel and the idle gap before it."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("dsdf::", "")) for r in rows))
# steps start at latent_renorm_kernel
starts = [i for i, e in enumerate(ev) if e[2].startswith("latent_renorm")]
steps = [ev[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = steps[-12:]  # timed region
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
wall = []
for s in steps:
    for i, (t0, t1, n) in enumerate(s):
        dur[(i, n)].append(t1 - t0)
        gap[(i, n)].append(t0 - s[i - 1][1] if i else 0)
for a, b in zip(steps[:-1], steps[1:]):
    wall.append(b[0][0] - a[0][0])
print("step wall (start-to-start) us: %.1f" % (sum(wall) / len(wall) / 1e3))
tg = 0
for k in sorted(dur):
    d, g = sum(dur[k]) / len(dur[k]) / 1e3, sum(gap[k]) / len(gap[k]) / 1e3
    tg += g
    print("%2d %-28s dur %8.1f us   gap before %6.1f us" % (k[0], k[1][:28], d, g))
print("sum of gaps inside a step: %.1f us" % tg)

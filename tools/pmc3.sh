export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $R/gpurun_out/pmc4 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc4.log 2>&1; echo "rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc5 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc5.log 2>&1; echo "rc=$?"
cd $R; python3 - <<'PY'
import csv, glob, collections
for d in ("pmc4", "pmc5"):
    f = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    if not f: print(d, "no file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        if any(k in n for k in ("fused", "dw_stream")):
            agg[n[:32]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, c in agg.items():
        print(d, n, {k: "%.4g" % (sum(v)/len(v)) for k, v in c.items()})
PY

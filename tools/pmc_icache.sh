# instruction-cache counters per kernel (separate pmc pass, kernel-trace only)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rm -rf $R/gpurun_out/pmc_ic
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH --output-format csv -d $R/gpurun_out/pmc_ic -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_ic.log 2>&1
echo "rc=$?"
python3 - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_ic/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][-32:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("%-34s" % k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY

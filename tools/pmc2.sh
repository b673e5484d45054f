export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/tools/lab_fused.py 16384 decode > $R/gpurun_out/pmc2.log 2>&1; echo "rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc3 -- python3 $R/tools/lab_fused.py 16384 decode > $R/gpurun_out/pmc3.log 2>&1; echo "rc=$?"
cd $R; python3 - <<'PY'
import csv, glob, collections
for d in ("pmc2", "pmc3"):
    f = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    if not f: print(d, "no file"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "fused_forward" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(d, k, "avg per launch = %.4g" % (sum(v)/len(v)), "n=", len(v))
PY

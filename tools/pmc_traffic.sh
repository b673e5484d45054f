# HBM traffic of the step's kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (MI355X_MICROARCH.md: TCC has 4
# slots, FETCH_SIZE costs 3, WRITE_SIZE 2).  Output: gpurun_out/pmc_traffic.json (copy to profiles/).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_fetch.log 2>&1; echo "rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc_write.log 2>&1; echo "rc=$?"
cd $R; python3 - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for d, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == key and "dsdf" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dsdf::", "").replace("(anonymous namespace)::", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k][key + "_KB_per_launch"] = sum(v) / len(v)
        out[k]["launches"] = len(v)
for k, v in out.items():
    f_, w_ = v.get("FETCH_SIZE_KB_per_launch", 0.0), v.get("WRITE_SIZE_KB_per_launch", 0.0)
    # gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM) -> doubled
    v["hbm_bytes_per_launch"] = (2.0 * f_ + w_) * 1024.0
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
    print(f"{k:28s} fetch {v.get('FETCH_SIZE_KB_per_launch',0)/1024:8.1f} MiB  write {v.get('WRITE_SIZE_KB_per_launch',0)/1024:8.1f} MiB  -> HBM {v['hbm_bytes_per_launch']/1e6:8.1f} MB/launch")
PY

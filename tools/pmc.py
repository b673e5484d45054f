"""rocprofv3 --pmc passes over bench.py and their reduction to per-kernel figures (measurement tooling, not product).

Used two ways:
  * by bench.py itself (rank 0, N = 1): the passes run as CHILD processes while the parent times the CPU baseline, so the
    bench line's roofline.traffic / executed_frac are measured in the same invocation;
  * stand-alone on the GPU box:   python tools/pmc.py <tag>    -> gpurun_out/<tag>_pmc_traffic.json, <tag>_mfma_util.json
    (copy them to profiles/).

Rules followed (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots): FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC has 4
slots: 3 + 2 do not fit); FETCH_SIZE x 2 on gfx950 (128-B requests tallied at 64 B); SQ counters (8 slots) + GRBM in one pass;
only --kernel-trace beside --pmc; the program itself (python3 bench.py ...) directly after `--`.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_F32 = 157.3e12
MFMA_COUNTERS = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_BF16",
                 "SQ_INSTS_MFMA", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]


def short(name):
    return name.split("(")[0].replace("void ", "").replace("dsdf::", "").replace("(anonymous namespace)::", "").strip()


def rocprof_path():
    return shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)


def run_pass(counters, bench_args, timeout=240, keep_dir=None, script="bench.py"):
    """One `rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py <bench_args>` run.
    Returns {kernel: {"counters": {name: avg per launch}, "launches": n, "duration_us": median}}."""
    rp = rocprof_path()
    if rp is None:
        raise RuntimeError("rocprofv3 not found")
    d = keep_dir or tempfile.mkdtemp(prefix="dsdf_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [rp, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "--", sys.executable,
           os.path.join(ROOT, script), *bench_args]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"rocprofv3 pass {counters} failed (rc {r.returncode}): {r.stderr[-400:]}")
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not cc:
        raise RuntimeError("no counter_collection.csv written")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(cc[0])):
        agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur = collections.defaultdict(list)
    if kt:
        for row in csv.DictReader(open(kt[0])):
            dur[short(row["Kernel_Name"])].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-3)
    out = {}
    for k, c in agg.items():
        if "kernel" not in k:
            continue
        e = dict(counters={n: sum(x) / len(x) for n, x in c.items()}, launches=max(len(x) for x in c.values()))
        if dur.get(k):
            xs = sorted(dur[k])
            e["duration_us"] = xs[len(xs) // 2]
        out[k] = e
    if keep_dir is None:
        shutil.rmtree(d, ignore_errors=True)
    return out


def traffic(bench_args, timeout=240):
    """HBM bytes per launch per kernel: FETCH_SIZE and WRITE_SIZE (KB) in separate passes, FETCH_SIZE doubled (gfx950)."""
    f = run_pass(["FETCH_SIZE"], bench_args, timeout)
    w = run_pass(["WRITE_SIZE"], bench_args, timeout)
    out = {}
    for k in set(f) | set(w):
        fk = f.get(k, {}).get("counters", {}).get("FETCH_SIZE", 0.0)
        wk = w.get(k, {}).get("counters", {}).get("WRITE_SIZE", 0.0)
        out[k] = dict(FETCH_SIZE_KB_per_launch=fk, WRITE_SIZE_KB_per_launch=wk, launches=f.get(k, w.get(k))["launches"],
                      hbm_bytes_per_launch=(2.0 * fk + wk) * 1024.0)
    return out


def mfma(bench_args, timeout=240, script="bench.py"):
    """Per kernel: executed fp32 MFMA FLOPs and MFMA-pipe busy share from the SQ counters of ONE pass.
      exec_mfma_flop  = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512   (one v_mfma_f32_32x32x2_f32 = 4096 FLOP; mops_per_mfma_inst reports
                                                              the measured MOPS per instruction so the unit can be checked: 8)
      mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): MFMA-pipe-busy share of the busy CUs' SIMD cycles
      clock_ghz       = GRBM_GUI_ACTIVE / 8 XCDs / duration
      exec_frac_of_peak = exec_mfma_flop / duration / 157.3 TFLOP/s  (durations of a PROFILED pass run a few % long)"""
    res = run_pass(MFMA_COUNTERS, bench_args, timeout, script=script)
    for k, e in res.items():
        v = e["counters"]
        mops, busy, cu = v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("SQ_BUSY_CU_CYCLES", 0.0)
        mops_bf = v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
        if v.get("SQ_INSTS_MFMA", 0.0) > 0:
            e["mops_per_mfma_inst"] = (mops + mops_bf) / v["SQ_INSTS_MFMA"]
            e["mfma_busy_cycles_per_inst"] = busy / v["SQ_INSTS_MFMA"]
        e["exec_mfma_flop"] = mops * 512.0
        e["exec_mfma_flop_bf16"] = mops_bf * 512.0     # (one v_mfma_f32_32x32x16_bf16 = 32768 FLOP = 64 MOPS)
        if cu > 0:
            e["mfma_busy_frac"] = busy / (4.0 * cu)
        t = e.get("duration_us", 0.0) * 1e-6
        if t > 0:
            e["exec_tflops"] = e["exec_mfma_flop"] / t / 1e12
            e["exec_frac_of_peak"] = e["exec_mfma_flop"] / t / PEAK_F32
            if v.get("GRBM_GUI_ACTIVE", 0.0) > 0:
                e["clock_ghz"] = v["GRBM_GUI_ACTIVE"] / 8.0 / t / 1e9
    return res


CHILD_ARGS = ["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-profile", "--no-pmc", "--no-extras"]

if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "--script":      # python tools/pmc.py --script tools/x.py [args]
    m = mfma(sys.argv[3:], script=sys.argv[2])
    for k, e in sorted(m.items(), key=lambda kv: -kv[1].get("duration_us", 0)):
        print(f"{k:28s} {e.get('duration_us', 0):8.1f} us  launches {e['launches']}  mfma_busy {100 * e.get('mfma_busy_frac', 0):5.1f} %  clock {e.get('clock_ghz', 0):.2f} GHz"
              f"  busy cyc/inst {e.get('mfma_busy_cycles_per_inst', 0):.1f}  counters {({c: round(v) for c, v in e['counters'].items()})}", flush=True)
elif __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "pmc"
    extra = sys.argv[2:]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    m = mfma(CHILD_ARGS + extra)
    json.dump(m, open(os.path.join(ROOT, "gpurun_out", f"{tag}_mfma_util.json"), "w"), indent=1)
    for k, e in sorted(m.items(), key=lambda kv: -kv[1].get("duration_us", 0)):
        print(f"{k:28s} {e.get('duration_us', 0):8.1f} us  exec {e.get('exec_tflops', 0):6.1f} TF ({100 * e.get('exec_frac_of_peak', 0):5.1f} %)"
              f"  mfma_busy {100 * e.get('mfma_busy_frac', 0):5.1f} %  clock {e.get('clock_ghz', 0):.2f} GHz"
              f"  mops/inst {e.get('mops_per_mfma_inst', 0):.2f}  busy cyc/inst {e.get('mfma_busy_cycles_per_inst', 0):.1f}", flush=True)
    t = traffic(CHILD_ARGS + extra)
    json.dump(t, open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_traffic.json"), "w"), indent=1)
    for k, v in sorted(t.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print(f"{k:28s} fetch {v['FETCH_SIZE_KB_per_launch'] / 1024:8.1f} MiB  write {v['WRITE_SIZE_KB_per_launch'] / 1024:8.1f} MiB  "
              f"-> HBM {v['hbm_bytes_per_launch'] / 1e6:8.1f} MB/launch")

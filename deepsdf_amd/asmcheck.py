"""Build-time check of the one place where inline asm leaves loads in flight across C++ code (dwstream.hpp dw_block_split).

The first asm statement there issues 16 ``ds_read_b128`` and ends in ``s_waitcnt lgkmcnt(8)``: the eight A fragments have
landed, the eight B reads are still in flight while the A terms are cut; a second statement (``s_waitcnt lgkmcnt(0)``) ends the
window.  The compiler cannot see loads inside asm, so BETWEEN the two statements it believes the B registers hold data: a copy,
an AGPR move or a spill of one of them placed there would read stale registers.  Whether it does that is a register-allocation
outcome -- so every build of the library (lab flags included) is checked on the gfx950 code object itself:

  for each run of 16 ds_read_b128 whose statement leaves reads in flight: no instruction up to the next ``s_waitcnt lgkmcnt(0)``
  may name a destination VGPR of those reads, and none may touch scratch (a spill in the window).

Tooling only (no GPU, no torch): ``python -m deepsdf_amd.asmcheck [lib]``; deepsdf_amd.build runs it after every compile.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "dw_stream_split_kernel"


class AsmHazard(RuntimeError):
    pass


def tools_available():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf", "llvm-objdump"))


def extract_code_object(lib, out_dir):
    fat, co = os.path.join(out_dir, "fat.bin"), os.path.join(out_dir, "co.elf")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
    return co


def disassemble(co, name_part):
    syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", co], capture_output=True, text=True, check=True).stdout
    names = sorted({f.split()[7] for f in syms.splitlines() if len(f.split()) >= 8 and f.split()[3] == "FUNC" and name_part in f.split()[7]})
    if len(names) != 1:
        raise AsmHazard(f"expected exactly one function matching {name_part!r} in the code object, found {names}")
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f"--disassemble-symbols={names[0]}", co],
                         capture_output=True, text=True, check=True).stdout
    ins = []
    for line in out.splitlines():
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if m:
            ins.append((m.group(1), m.group(2)))
    return ins


def vgprs(operands):
    """Set of VGPR numbers an operand string names: v7, v[4:7] (AGPRs a.. are a different file and do not count)."""
    regs = set()
    for m in re.finditer(r"(?<![a-z_0-9])v(\d+)(?![\d:\]])", operands):
        regs.add(int(m.group(1)))
    for m in re.finditer(r"(?<![a-z_0-9])v\[(\d+):(\d+)\]", operands):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return regs


def check_split_wait_windows(ins):
    """Returns the number of windows checked; raises AsmHazard on a violation.  A window opens behind a run of 16 ds_read_b128
    (the fragment reads of dwstream.hpp) that is not closed by an ``s_waitcnt lgkmcnt(0)`` at once, and extends to the next
    ``s_waitcnt lgkmcnt(0)``: directly behind the reads either ``lgkmcnt(8)`` (dw_block_split_v1: the first eight have landed, the
    last eight are in flight) or nothing (dw_block_split: all sixteen are in flight)."""
    windows = 0
    for i, (op, args) in enumerate(ins):
        if i < 16 or ins[i - 1][0] != "ds_read_b128" or op == "ds_read_b128":
            continue
        reads = ins[i - 16:i]
        if any(o != "ds_read_b128" for o, _ in reads):
            continue
        if op == "s_waitcnt" and "lgkmcnt(0)" in args:
            continue                       # reads and their wait in one statement: no window
        first_open = 8 if (op == "s_waitcnt" and "lgkmcnt(8)" in args) else 0
        in_flight = set()
        for _, a in reads[first_open:]:
            in_flight |= vgprs(a.split(",")[0])
        if len(in_flight) != 4 * (16 - first_open):
            raise AsmHazard(f"window at instruction {i}: the reads in flight name {len(in_flight)} destination VGPRs, "
                            f"expected {4 * (16 - first_open)}")
        start = i + 1 if first_open else i
        for j in range(start, len(ins)):
            o, a = ins[j]
            if o == "s_waitcnt" and "lgkmcnt(0)" in a:
                break
            if o.startswith("scratch_") or (o.startswith("buffer_") and "off" in a and "s[0:3]" in a):
                raise AsmHazard(f"scratch traffic inside the split-wait window: {o} {a}")
            hit = vgprs(a) & in_flight
            if hit:
                raise AsmHazard(f"instruction {j} inside the split-wait window names registers whose reads are still in flight "
                                f"(v{sorted(hit)}): {o} {a}")
            if o in ("s_endpgm", "s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz",
                     "s_cbranch_execz", "s_cbranch_execnz", "s_setpc_b64"):
                raise AsmHazard(f"control flow ({o}) inside the split-wait window before its s_waitcnt lgkmcnt(0)")
        else:
            raise AsmHazard("split-wait window is never closed by s_waitcnt lgkmcnt(0)")
        windows += 1
    return windows


def check_library(lib, expect_windows=True):
    """Raises AsmHazard if the library's dw_stream_split_kernel violates the rule; returns the number of windows found
    (0 is an error unless the build was made with the single-wait variant, expect_windows=False)."""
    with tempfile.TemporaryDirectory(prefix="dsdf_asmcheck_") as d:
        ins = disassemble(extract_code_object(lib, d), KERNEL)
    n = check_split_wait_windows(ins)
    if expect_windows and n == 0:
        raise AsmHazard(f"{KERNEL}: no split-wait window found (16 ds_read_b128 left in flight); the check would be vacuous")
    return n


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdsdf_hip.so")
    print(f"{KERNEL}: {check_library(lib, expect_windows='--allow-none' not in sys.argv)} split-wait window(s) clean")


# ---- second audit: MFMA source registers overwritten by a later load (fused_bf16x8.hpp, two waves per SIMD) -----------------
# Measured on MI355X (profiles/r02_lab_bf16x8_race.log): with two waves per SIMD, a ds_read / buffer_load that the register
# allocator gave the registers an MFMA issued JUST BEFORE it reads as srcA / srcB produced run-to-run differences in 3 % of the
# rows.  The k-loop was rewritten so that a step's loads only replace what the PREVIOUS step consumed and are issued after the
# step's own MFMAs.  That is a property of the emitted code, so it is measured on the code object: for every load with a VGPR
# destination, the number of OTHER MFMAs issued between the last MFMA that reads one of those registers as srcA / srcB and the
# load, over every path (loop back-edges included).
def disassemble_with_addresses(co, name_part):
    syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", co], capture_output=True, text=True, check=True).stdout
    names = sorted({f.split()[7] for f in syms.splitlines() if len(f.split()) >= 8 and f.split()[3] == "FUNC" and name_part in f.split()[7]})
    if len(names) != 1:
        raise AsmHazard(f"expected exactly one function matching {name_part!r} in the code object, found {names}")
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f"--disassemble-symbols={names[0]}", co],
                         capture_output=True, text=True, check=True).stdout
    base, ins = None, []
    for line in out.splitlines():
        m = re.match(r"^([0-9a-f]+) <", line)
        if m and base is None:
            base = int(m.group(1), 16)
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
        if m:
            tgt = re.search(r"<[^>]*\+0x([0-9a-f]+)>", m.group(4))
            ins.append(dict(op=m.group(1), args=m.group(2), addr=int(m.group(3), 16),
                            target=(base + int(tgt.group(1), 16)) if tgt and base is not None else None))
    return ins


def mfma_src_reuse_distances(ins, limit=12, fall_through_only=False, edges=None):
    """{load index: MFMAs issued strictly between the last MFMA reading one of the load's destination VGPRs as srcA/srcB and
    the load} for every ds_read* / buffer_load* / global_load* with a VGPR destination that has such a reader within `limit`
    MFMAs on some path.  Which control-flow edges a path may follow (`edges`):
      "all"           fall-through and every branch edge: also contains paths the program's own guards exclude (a forward branch
                      over a block of MFMAs taken together with the branch that only a step WITH MFMAs takes);
      "loops"         fall-through and BACKWARD branch edges (target <= branch: loop back-edges) -- every conditional forward
                      branch not taken, every loop iterated: the steady state of a k-loop, where the loads at the top of the body
                      follow the last MFMAs of the previous trip;
      "fall_through"  fall-through edges alone (every conditional branch not taken; `fall_through_only=True` is the old name)."""
    if edges is None:
        edges = "fall_through" if fall_through_only else "all"
    if edges not in ("all", "loops", "fall_through"):
        raise ValueError(edges)
    by_addr = {x["addr"]: i for i, x in enumerate(ins)}
    preds = [[] for _ in ins]
    for i, x in enumerate(ins):
        uncond = x["op"] in ("s_branch", "s_endpgm", "s_setpc_b64")
        if i + 1 < len(ins) and not uncond:
            preds[i + 1].append(i)
        if edges != "fall_through" and x["op"].startswith(("s_branch", "s_cbranch")) and x["target"] in by_addr:
            if edges == "all" or x["target"] <= x["addr"]:
                preds[by_addr[x["target"]]].append(i)
    src_ab, acc_regs = {}, set()
    for i, x in enumerate(ins):
        if x["op"].startswith("v_mfma"):
            ops = [o.strip() for o in x["args"].split(",")]
            src_ab[i] = vgprs(ops[1]) | vgprs(ops[2])
            acc_regs |= vgprs(ops[0])
    out = {}
    for i, x in enumerate(ins):
        if not x["op"].startswith(("ds_read", "buffer_load", "global_load")) or " lds" in (" " + x["args"]):
            continue
        dest = vgprs(x["args"].split(",")[0])
        if not dest:
            continue
        best, seen, stack = None, set(), [(p, 0) for p in preds[i]]
        while stack:
            j, n = stack.pop()
            if (j, n) in seen or n > limit:
                continue
            seen.add((j, n))
            if j in src_ab:
                if src_ab[j] & dest:
                    best = n if best is None else min(best, n)
                    continue
                n += 1
            elif ins[j]["op"].startswith("v_") and vgprs(",".join(ins[j]["args"].split(",")[1:])) & acc_regs:
                continue          # a VALU instruction READING an MFMA result (epilogue arithmetic, bf8_drain's v_readfirstlane): it
                                  # completes only after the MFMA that writes the register, and the matrix pipe is in-order
            elif vgprs(ins[j]["args"].split(",")[0]) & dest and not ins[j]["op"].startswith(("v_cmp", "s_", "ds_write", "buffer_store", "global_store")):
                continue          # an earlier WRITER of the same register on this path: what it overwrote is no longer the MFMA's operand
            stack.extend((p, n) for p in preds[j])
        if best is not None:
            out[i] = best
    return out


def check_mfma_src_reuse(lib, kernel="fused_forward_bf16x8_kernel", min_distance=1):
    """Raises AsmHazard if, on a path of fall-through and loop back-edges (every loop iterated, no forward branch taken: the
    k-loop's steady state included), some load in `kernel` overwrites srcA/srcB registers of an MFMA with fewer than `min_distance`
    other MFMAs issued in between (the signature of the first 8-wave k-loop: 46 such loads; the rewritten loop: none);
    returns (smallest distance found, number of load/MFMA pairs looked at)."""
    with tempfile.TemporaryDirectory(prefix="dsdf_asmcheck_") as d:
        ins = disassemble_with_addresses(extract_code_object(lib, d), kernel)
    dist = mfma_src_reuse_distances(ins, edges="loops")
    if not dist:
        return None, 0
    worst = min(dist.values())
    if worst < min_distance:
        bad = [f"{ins[i]['op']} {ins[i]['args']} @ {ins[i]['addr']:#x} (distance {n})" for i, n in dist.items() if n < min_distance]
        raise AsmHazard(f"{kernel}: {len(bad)} load(s) overwrite MFMA source registers after fewer than {min_distance} other MFMAs: "
                        + "; ".join(bad[:4]))
    return worst, len(dist)

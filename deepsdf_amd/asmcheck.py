"""Build-time check of the one place where inline asm leaves loads in flight across C++ code (dwstream.hpp dw_block_split).

The first asm statement there issues 16 ``ds_read_b128`` and ends in ``s_waitcnt lgkmcnt(8)``: the eight A fragments have
landed, the eight B reads are still in flight while the A terms are cut; a second statement (``s_waitcnt lgkmcnt(0)``) ends the
window.  The compiler cannot see loads inside asm, so BETWEEN the two statements it believes the B registers hold data: a copy,
an AGPR move or a spill of one of them placed there would read stale registers.  Whether it does that is a register-allocation
outcome -- so every build of the library (lab flags included) is checked on the gfx950 code object itself:

  for each ``s_waitcnt lgkmcnt(8)`` preceded by 16 ds_read_b128: no instruction up to the next ``s_waitcnt lgkmcnt(0)`` may
  name a destination VGPR of the last eight reads, and none may touch scratch (a spill in the window).

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
    """Returns the number of windows checked; raises AsmHazard on a violation."""
    windows = 0
    for i, (op, args) in enumerate(ins):
        if op != "s_waitcnt" or "lgkmcnt(8)" not in args:
            continue
        reads = ins[i - 16:i]
        if len(reads) != 16 or any(o != "ds_read_b128" for o, _ in reads):
            continue                       # some other lgkmcnt(8) of the compiler's own: not the asm window
        in_flight = set()
        for _, a in reads[8:]:
            in_flight |= vgprs(a.split(",")[0])
        if len(in_flight) != 32:
            raise AsmHazard(f"window at instruction {i}: the eight B reads name {len(in_flight)} destination VGPRs, expected 32")
        for j in range(i + 1, len(ins)):
            o, a = ins[j]
            if o == "s_waitcnt" and "lgkmcnt(0)" in a:
                break
            if o.startswith("scratch_") or (o.startswith("buffer_") and "off" in a and "s[0:3]" in a):
                raise AsmHazard(f"scratch traffic inside the split-wait window: {o} {a}")
            hit = vgprs(a) & in_flight
            if hit:
                raise AsmHazard(f"instruction {j} inside the split-wait window names B registers still in flight "
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
        raise AsmHazard(f"{KERNEL}: no split-wait window found (16 ds_read_b128 + s_waitcnt lgkmcnt(8)); the check would be vacuous")
    return n


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdsdf_hip.so")
    print(f"{KERNEL}: {check_library(lib, expect_windows='--allow-none' not in sys.argv)} split-wait window(s) clean")

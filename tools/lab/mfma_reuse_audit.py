"""Lab: MFMA source-register reuse distances of a kernel in one or more library builds (deepsdf_amd/asmcheck.py).
  python tools/lab/mfma_reuse_audit.py [kernel-name-part] lib1.so [lib2.so ...]"""
import collections
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepsdf_amd import asmcheck  # noqa: E402


def audit(lib, kernel):
    with tempfile.TemporaryDirectory() as d:
        ins = asmcheck.disassemble_with_addresses(asmcheck.extract_code_object(lib, d), kernel)
    dist = asmcheck.mfma_src_reuse_distances(ins)
    ft = asmcheck.mfma_src_reuse_distances(ins, fall_through_only=True)
    kinds = collections.Counter()
    for i, n in dist.items():
        if n != 0:
            continue
        dest = asmcheck.vgprs(ins[i]["args"].split(",")[0])
        j, hit = i - 1, False
        while j >= 0 and not ins[j]["op"].startswith(("s_cbranch", "s_branch")):     # straight back, no branch crossed
            if ins[j]["op"].startswith("v_mfma"):
                ops = [o.strip() for o in ins[j]["args"].split(",")]
                hit = bool((asmcheck.vgprs(ops[1]) | asmcheck.vgprs(ops[2])) & dest)
                break
            j -= 1
        kinds[(ins[i]["op"], "straight-line" if hit else "only across a branch")] += 1
    return len(ins), sum(1 for x in ins if x["op"].startswith("v_mfma")), collections.Counter(dist.values()), kinds, collections.Counter(ft.values())


if __name__ == "__main__":
    args = sys.argv[1:]
    kernel = args.pop(0) if args and not args[0].endswith(".so") else "fused_forward_bf16x8_kernel"
    for lib in args:
        n, nm, c, k, ft = audit(lib, kernel)
        print(f"{lib}: {kernel}: {n} instructions, {nm} MFMAs; loads whose destination an MFMA reads as srcA/srcB, by the number of "
              f"other MFMAs issued in between: {sorted(c.items())}")
        print("   on fall-through paths only:", sorted(ft.items()))
        print("   distance 0, by load kind and path:", {f"{a} / {b}": v for (a, b), v in sorted(k.items())})

"""Build libdsdf_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  No torch involved."""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "dsdf_api.hip")
# every kernel source is a dependency: a stale .so would travel to the GPU box and be the thing measured
DEPS = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")) + glob.glob(os.path.join(HERE, "csrc", "*.hpp"))) + [
    os.path.join(os.path.dirname(HERE), "include", "dsdf.h")]
LIB = os.environ.get("DSDF_LIB_PATH") or os.path.join(HERE, "libdsdf_hip.so")   # override: lab builds only (tools/lab_*.sh build those
# with hipcc directly: they are NOT audited by asmcheck unless built through build_library -- measurement libraries, never shipped)


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_library(force=False, verbose=False):
    """Compile csrc/dsdf_api.hip -> deepsdf_amd/libdsdf_hip.so for gfx950.  Returns the library path."""
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC]
    cmd += os.environ.get("DSDF_HIPCC_FLAGS", "").split()      # lab builds only (-D switches of tools/lab_*.sh)
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    # the one inline-asm window that keeps loads in flight across compiler-scheduled code is verified on the code object of
    # EVERY build, lab flags included (asmcheck.py); a library that violates it -- or that could not be checked -- is removed,
    # not shipped: left on disk it would pass is_stale() next time and be loaded unverified
    from . import asmcheck
    if asmcheck.tools_available():
        try:
            asmcheck.check_library(LIB, expect_windows="-DDW_SPLIT_ONE_WAIT=1" not in cmd)
            asmcheck.check_mfma_src_reuse(LIB, min_distance=2)      # fused_bf16x8.hpp: a step is >= 2 MFMAs (one n-tile per wave)
        except BaseException:      # AsmHazard, a failing llvm tool (CalledProcessError), an interrupt: no unverified library stays
            if os.path.exists(LIB):
                os.remove(LIB)
            raise
    else:
        import warnings
        warnings.warn("deepsdf_amd.build: ROCm LLVM tools not found under " + asmcheck.LLVM + " -- the inline-asm audits of the code "
                      "object (asmcheck.py) were SKIPPED for " + LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))

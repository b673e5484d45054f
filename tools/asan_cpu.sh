# Host-side AddressSanitizer run (CPU only: GPU ASan / xnack builds are not available on the pool).  The C ABI's host code --
# validation, workspace planning, parameter layouts, launch-argument assembly up to the first device call -- is compiled with
# -fsanitize=address (device code unsanitised and at the shipped -O3: the build's asm audits apply to this library too, and refuse what
# -O1 makes of the inline-asm windows) and the ABI tests run against that library.  Run in the authoring container:
#     bash tools/asan_cpu.sh
set -e
cd "$(dirname "$0")/.."
ASAN_LIB=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
DSDF_LIB_PATH=/tmp/libdsdf_asan.so DSDF_HIPCC_FLAGS="-fsanitize=address -fno-gpu-sanitize -g -Xarch_host -O1" python -m deepsdf_amd.build > /dev/null
DSDF_LIB_PATH=/tmp/libdsdf_asan.so LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 \
  python -m pytest tests/test_abi_cpu.py -q

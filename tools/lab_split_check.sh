# lab: gemm_split parity (the gemm_split, shipped-shape and two-phase tests) + step / kernel times per library variant under tools/lab/variants/
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -s -k "gemm_split or shipped_experiment or phased" > gpurun_out/r3_t4.log 2>&1
rc=$?; grep -E "gemm_split|passed|failed|Error|error" gpurun_out/r3_t4.log | tail -30
[ $rc -eq 0 ] || { echo "FAILED: pytest rc $rc"; exit 1; }
for round in 1 2; do bash tools/lab_split.sh; done 2>&1 | tee gpurun_out/r3_asm_ab.log

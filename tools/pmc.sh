export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
grep -i -E "MFMA|GRBM_GUI|SQ_BUSY_CYC|SQ_WAVE_CYCLES|SQ_WAIT|SQ_ACTIVE_INST|LDS_BANK|SQ_INSTS_VALU " $R/gpurun_out/counters.txt | head -60
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc1.log 2>&1; echo "pmc rc=$?"; tail -3 $R/gpurun_out/pmc1.log; ls -R $R/gpurun_out/pmc1 | head

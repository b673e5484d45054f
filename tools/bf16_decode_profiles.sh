# judged artefacts of the config-5 inference forward (dsdf_decode_latent, 16384 and 1 M points): rocprofv3 kernel stats, MFMA counters,
# per-layer stamps.  usage (GPU box): bash tools/bf16_decode_profiles.sh r02
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=${1:-r02}
cd /tmp && rm -rf $R/gpurun_out/dstats && LAB_ONLY_BF16=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dstats -- python3 $R/tools/lab_bf16_fwd.py 16384 > $R/gpurun_out/${T}_bf16_decode.log 2>&1
cp $(find $R/gpurun_out/dstats -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${T}_bf16_decode_kernel_stats.csv
cd $R && LAB_ONLY_BF16=1 python3 tools/lab_bf16_fwd.py 1048576 2>/dev/null | tail -n 1 >> gpurun_out/${T}_bf16_decode.log
cd $R && LAB_ONLY_BF16=1 python3 tools/pmc.py --script tools/lab_bf16_fwd.py 16384 2>&1 | grep -v "^decode" | head -n 4 > gpurun_out/${T}_bf16_decode_mfma.log
cat gpurun_out/${T}_bf16_decode.log | tail -n 3; head -n 4 gpurun_out/${T}_bf16_decode_kernel_stats.csv | cut -d, -f1-8; cut -c1-400 gpurun_out/${T}_bf16_decode_mfma.log

#!/bin/bash
# rocprofv3 kernel trace of FRI commits at 2^$1 -> gpurun_out/trace_$1/ ; timeline of the last commit -> gpurun_out/timeline_$1.txt
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
LOG=${1:-20}
rm -rf gpurun_out/trace_$LOG
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_$LOG -o t -- python3 tools/bench_commit.py stark_anatomy_amd/libstark_hip.so $LOG > gpurun_out/trace_$LOG.log 2>&1 || exit 1
CSV=$(find gpurun_out/trace_$LOG -name '*kernel_trace.csv' | head -1)
python3 tools/trace_summary.py "$CSV" k_leaf_subtree${2:-} 10 > gpurun_out/timeline_$LOG.txt 2>&1

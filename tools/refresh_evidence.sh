#!/bin/bash
# Regenerates the measurement evidence of a round on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of bench.py, kernel traces of three workloads (LDE + commit at 2^24 and 2^20, 128 x 2^14 batch),
#   four separate --pmc passes per workload (FETCH_SIZE | WRITE_SIZE | SQ issue/wait | SQ instruction mix, LDS).
# Everything lands in gpurun_out/evidence/; tools/evidence_table.py turns it into the tables under profiles/.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
export STARK_HIP_CHAIN_SPIN_US=200000   # the profiler slows the host side: do not let wait kernels give up and re-launch rounds while it watches
OUT=gpurun_out/evidence
rm -rf $OUT && mkdir -p $OUT
run() { echo "== $*" >> $OUT/log.txt; "$@" >> $OUT/log.txt 2>&1; }
run timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o b -- python3 bench.py --steps 5 --warmup 2 --no-cpu || exit 1
for W in "24 1" "20 1" "14 128"; do
    set -- $W; TAG=w$1x$2
    run timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$TAG -o t -- python3 tools/pmc_workload.py $1 $2 || exit 1
    run timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${TAG}_fetch -o p -- python3 tools/pmc_workload.py $1 $2 || exit 1
    run timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${TAG}_write -o p -- python3 tools/pmc_workload.py $1 $2 || exit 1
    run timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${TAG}_sq -o p -- python3 tools/pmc_workload.py $1 $2 || exit 1
    run timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_${TAG}_mix -o p -- python3 tools/pmc_workload.py $1 $2 || exit 1
done
find $OUT -name '*.csv' | sort > $OUT/files.txt
echo done >> $OUT/log.txt

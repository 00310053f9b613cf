#!/bin/bash
# Development aid: where the waves of tools/quick_bench.py (metric shape) wait -- LDS / vector-memory / instruction-fetch
# latencies (accumulated in-flight level over issued count) and full-FIFO cycles, in separate passes.
# usage: [BENCH=tools/quick_bench_f32.py] bash tools/pmc_latency.sh <tag>
set -o pipefail
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
out=gpurun_out/pmcl_$tag
rm -rf $out && mkdir -p $out
pass() {
    local target=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_run -- python3 ${BENCH:-tools/quick_bench.py} ${CHANNELS:-1024} > /dev/null || return 1
    find $out/pmc_run -name "*counter_collection.csv" | head -1 | xargs -I{} python3 tools/pmc_summary.py {} >> $out/$target
    rm -rf $out/pmc_run
}
pass lat.txt SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE || exit 1
pass lat.txt SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE || exit 1
pass lat.txt SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM GRBM_GUI_ACTIVE || exit 1
cat $out/lat.txt

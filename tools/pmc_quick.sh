#!/bin/bash
# Development aid: hardware counters of tools/quick_bench.py (metric shape) in separate passes.
# usage: [BENCH=tools/quick_bench_f32.py] bash tools/pmc_quick.sh <tag>   (environment, e.g. VOC_FAST16 / VOC_LIB, is passed through)
set -o pipefail
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
out=gpurun_out/pmcq_$tag
rm -rf $out && mkdir -p $out
pass() {
    local target=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_run -- python3 ${BENCH:-tools/quick_bench.py} ${CHANNELS:-1024} > /dev/null || return 1
    find $out/pmc_run -name "*counter_collection.csv" | head -1 | xargs -I{} python3 tools/pmc_summary.py {} >> $out/$target
    rm -rf $out/pmc_run
}
pass sq.txt SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE || exit 1
pass sq.txt SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE || exit 1
cat $out/sq.txt

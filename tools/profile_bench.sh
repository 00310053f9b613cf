#!/bin/bash
# Collects the profile evidence of bench.py on the GPU box: bench line, rocprofv3 kernel statistics of the
# same command, and hardware counters in separate passes (one counter group per run).  Results land
# under gpurun_out/profile/ as small text / csv files.   usage: bash tools/profile_bench.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
out=gpurun_out/profile
rm -rf $out && mkdir -p $out
timeout -k 10 400 python3 bench.py > $out/bench.json || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --no-cpu-baseline --no-parity > $out/bench_under_rocprof.json || exit 1
find $out/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/kt
pass() {  # target file, counters...
    local target=$1; shift
    timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_run -- python3 bench.py --no-cpu-baseline --no-parity --steps 3 --warmup 1 > /dev/null || return 1
    find $out/pmc_run -name "*counter_collection.csv" | head -1 | xargs -I{} python3 tools/pmc_summary.py {} --json $out/pmc.json >> $out/$target
    rm -rf $out/pmc_run
}
pass pmc_sq.txt SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE || exit 1
pass pmc_sq.txt SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE || exit 1
pass pmc_hbm.txt FETCH_SIZE || exit 1
pass pmc_hbm.txt WRITE_SIZE || exit 1
python3 tools/pmc_to_json.py $out/pmc.json --out $out || exit 1
ls -la $out

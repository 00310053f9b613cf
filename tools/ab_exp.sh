#!/bin/bash
# Development aid: times experimental builds (exp/<name>/lib.so, tools/exp_build.sh) back to back on one box.
# usage: bash tools/ab_exp.sh name1 name2 ...      (CHANNELS, default 1024)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
for rep in 1 2; do
for n in "$@"; do
    echo "== $n"
    VOC_LIB=exp/$n/lib.so timeout -k 10 200 python3 tools/quick_bench.py ${CHANNELS:-1024} 2>&1 | grep -v amdgpu.ids || exit 1
done
done

#!/bin/bash
# Development aid: N = 8192 round trip, eight against sixteen points per thread, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
for rep in 1 2; do for v in 0 1; do echo "== VOC_FAST16=$v"; VOC_FAST16=$v VOC_LIB=exp/n8k/lib.so timeout -k 10 200 python3 tools/quick_bench.py 512 8192 2048 2>&1 | grep -v amdgpu.ids || exit 1; done; done

#!/bin/bash
# Development aid: the metric shape with eight / sixteen points per thread in analysis and synthesis, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
for rep in 1 2; do for v in "0 0" "1 0" "1 1"; do set -- $v; echo "== ANA16=$1 SYN16=$2"; VOC_FAST16_ANA=$1 VOC_FAST16_SYN=$2 timeout -k 10 200 python3 tools/quick_bench.py ${CHANNELS:-1024} ${NFFT:-4096} 2>&1 | grep -v amdgpu.ids || exit 1; done; done

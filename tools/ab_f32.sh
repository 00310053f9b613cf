#!/bin/bash
# Development aid: single-precision round trip, builds / kernel choices back to back on one box.
# usage: bash tools/ab_f32.sh "lib ANA SYN [CHANNELS N]" ...   (lib: a .so path or "-" for the product library;
#        ANA / SYN = 0 | 1: sixteen points per thread)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
for rep in 1 2; do
for spec in "$@"; do
    read -r lib ana syn ch n <<< "$spec"
    echo "== $lib ana16=$ana syn16=$syn"
    [ "$lib" = "-" ] && lib=""
    VOC_LIB=$lib VOC_FAST16_F32_ANA=$ana VOC_FAST16_F32_SYN=$syn timeout -k 10 200 python3 tools/quick_bench_f32.py ${ch:-1024} ${n:-4096} 2>&1 | grep -v amdgpu.ids || exit 1
done
done

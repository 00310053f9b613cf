#!/bin/bash
# Development aid: builds an experimental copy of the library (N = 4096 kernels only, seconds) as
# exp/<name>/lib.so with its assembly (dev.s) and register report beside it.
# usage: tools/exp_build.sh <name> [-DMACRO=value ...]      then: VOC_LIB=exp/<name>/lib.so python tools/quick_bench.py 1024
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/exp/$name"
cd "$root/exp/$name" || exit 1
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden \
    -Wall -Wno-unused-result -DVOC_DEV_ONLY_LOG2M=${LOG2M:-11} -save-temps=obj -Rpass-analysis=kernel-resource-usage "$@" \
    -o "$root/exp/$name/lib.so" "$root/vocoder_amd/csrc/vocoder_hip.cpp" 2> remarks.txt || { tail -20 remarks.txt; exit 1; }
mv vocoder_hip-hip-amdgcn-amd-amdhsa-gfx950.s dev.s; rm -f vocoder_hip*
grep -A12 "Function Name: _ZN3voc2[01]\(analysis\|synthesis\)_fast_kernelILi${LOG2M:-11}ELb1ELi2EdL\?b\?0\?E" remarks.txt | grep -o "Function Name: _ZN3voc2[01][a-z]*\|VGPRs: [0-9]*\|VGPRs Spill: [0-9]*\|ScratchSize.*: [0-9]*" | tr '\n' ' '
echo

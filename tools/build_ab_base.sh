#!/bin/bash
# tools/build_ab_base.sh [commit] : the library as of `commit` (default HEAD) into ab/lib_base.so, for A/B runs against the working
# tree's build on one GPU box (tools/ab_graph.sh; SHN_HIP_LIB picks the build).  ab/ is git-ignored and travels with gpurun.
set -e
cd "$(dirname "$0")/.."
C=${1:-HEAD}
W=/tmp/ab_base_src; rm -rf $W; mkdir -p $W/obj ab
git archive $C shannon_amd/csrc include | tar -x -C $W
ls $W/shannon_amd/csrc/*.hip | xargs -P 6 -I{} sh -c '/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -w -c {} -o '$W'/obj/$(basename {} .hip).o'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib_base.so $W/obj/*.o
ls -la ab/lib_base.so

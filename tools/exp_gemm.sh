#!/bin/bash
# kernel experiments on the standalone SYRK/GEMM bench
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/gemm_bench.bin tools/gemm_bench.cpp -Lcapital_amd/lib -lcapital_amd -Wl,-rpath,$R/capital_amd/lib 2>/dev/null
run() { echo "== $*"; env "$@" tools/gemm_bench.bin 32768 32768 1024 1 4; env "$@" tools/gemm_bench.bin 32768 32768 512 1 4; env "$@" tools/gemm_bench.bin 8192 8192 8192 0 3; env "$@" tools/gemm_bench.bin 16384 16384 1024 1 6; }
for v in "$@"; do run $v; done

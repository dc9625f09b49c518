#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/gemm_bench.bin tools/gemm_bench.cpp -Lcapital_amd/lib -lcapital_amd -Wl,-rpath,$R/capital_amd/lib 2>/dev/null
for d in 0 1 2 3 0; do echo "== CAP_DIAG=$d"; CAP_DIAG=$d tools/gemm_bench.bin 8192 8192 8192 0 3; done

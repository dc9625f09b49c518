#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/contend.bin tools/contend.cpp -Lcapital_amd/lib -lcapital_amd -Wl,-rpath,$R/capital_amd/lib 2>&1 | grep -E "error" 
tools/contend.bin h l; tools/contend.bin n n; tools/contend.bin h n; tools/contend.bin l l

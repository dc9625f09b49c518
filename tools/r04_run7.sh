set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MP_RESERVE=0,8,16,32 timeout 600 python tools/mp_kernel_ab.py 65536 8 > gpurun_out/r04_mp_ab3.log 2>&1; tail -20 gpurun_out/r04_mp_ab3.log

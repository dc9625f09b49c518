set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MP_MIN_TILES=4096,8192,16384 timeout 600 python tools/mp_kernel_ab.py 65536 8 2 2>&1 | grep -v amdgpu > gpurun_out/r04_mp_ab6.log; cat gpurun_out/r04_mp_ab6.log

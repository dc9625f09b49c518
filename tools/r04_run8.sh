set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MP_MIN_TILES=8000,16000,24000,32000,48000 timeout 600 python tools/mp_kernel_ab.py 65536 8 > gpurun_out/r04_mp_ab4.log 2>&1; tail -14 gpurun_out/r04_mp_ab4.log

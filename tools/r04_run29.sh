set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r04_bench_final.log 2> gpurun_out/r04_bench_final.err
grep '^{' gpurun_out/r04_bench_final.log | cut -c1-300
tail -3 gpurun_out/r04_bench_final.err

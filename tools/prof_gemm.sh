#!/bin/bash
# rocprofv3 passes on the standalone GEMM/SYRK bench; outputs under gpurun_out/prof_gemm (csv)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_gemm; rm -rf $OUT; mkdir -p $OUT
cd /tmp
ARGS="${GEMM_ARGS:-32768 32768 512 1 3}"
run() { name=$1; shift; timeout 120 rocprofv3 "$@" --output-format csv -d $OUT/$name -o gemm -- $R/tools/gemm_bench.bin $ARGS > $OUT/$name.log 2>&1; tail -n 1 $OUT/$name.log; }
run trace --kernel-trace --stats
run pmc1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES
run pmc2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE
run pmc3 --pmc FETCH_SIZE
run pmc4 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
find $OUT -name "*.csv" | head -30

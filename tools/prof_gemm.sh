#!/bin/bash
# rocprofv3 passes on the standalone GEMM/SYRK bench; outputs under gpurun_out/prof_gemm
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_gemm; mkdir -p $OUT
cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -ciE "mfma" $OUT/counters.txt
ARGS="${GEMM_ARGS:-32768 32768 512 1 3}"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o gemm -- $R/tools/gemm_bench.bin $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F64 SQ_WAVES -d $OUT/pmc1 -o gemm -- $R/tools/gemm_bench.bin $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc2 -o gemm -- $R/tools/gemm_bench.bin $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc3 -o gemm -- $R/tools/gemm_bench.bin $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCP_TCC_READ_REQ_sum -d $OUT/pmc4 -o gemm -- $R/tools/gemm_bench.bin $ARGS > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -30

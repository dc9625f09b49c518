#!/bin/bash
# Experiment helper: link one libcapital_amd variant per value of a -D switch of ONE source file (other objects reused).
#   tools/variant_libs.sh cqr_kernels.hip CQY_LATE 9 4 2 0   ->  gpurun_variants/libcapital_amd.CQY_LATE_<v>.so
set -e
src=$1; macro=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd); OBJ=$R/capital_amd/lib/obj; OUT=$R/gpurun_variants; mkdir -p $OUT
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -D$macro=$v -I$R/include -c $R/capital_amd/csrc/$src -o $OUT/$src.$v.o
  objs=$(ls $OBJ/*.o | grep -v "/$src.o"); 
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libcapital_amd.${macro}_$v.so $objs $OUT/$src.$v.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  rm $OUT/$src.$v.o
done
ls -la $OUT

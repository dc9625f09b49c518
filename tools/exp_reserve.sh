#!/bin/bash
# tail-only CU reservation for the diagonal-block chain (options reserve / reserve_m): A/B at N = 32768 and 65536
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
B=tools/opt_bench.bin
timeout 300 $B 32768 -1 3 -- -- reserve=8 reserve_m=16384 -- reserve=16 reserve_m=16384 -- reserve=32 reserve_m=16384 \
  -- reserve=16 reserve_m=16384 occ1_m=0 -- reserve=16 reserve_m=24576 -- reserve=16 reserve_m=8192 -- reserve=16 -- 
timeout 300 $B 65536 -1 2 -- -- reserve=16 reserve_m=16384

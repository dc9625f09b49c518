#!/bin/bash
# The 4-rank "hard" distributed mixed-precision case failed once INSIDE the batch launch of the suite (the one-rank cross-check factor
# 29 % off) and never in 24 fresh-process runs nor in 36 cases of the three mixed cases alone in one launch: repeat the WHOLE case list of
# that launch (tools/r05_p4_cases.txt, from the failing log) REPS times in one launch, with the worker's per-block diagnostics, carrying on
# after a failure of the cross-check (CAP_TEST_DIAG_CONTINUE).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
REPS=${REPS:-3}
python - <<PY
import json
lines = [l.split() for l in open("tools/r05_p4_cases.txt") if l.strip()]
cases = []
for i in range($REPS):
    for argv in lines:
        cases.append({"id": "rep%d %s" % (i, " ".join(argv)), "argv": argv})
json.dump(cases, open("/tmp/flake_cases.json", "w"))
print(len(cases), "cases")
PY
env CAP_TEST_DIAG_CONTINUE=1 OMP_NUM_THREADS=8 MASTER_ADDR=127.0.0.1 timeout ${LIMIT:-400} python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 \
    --master-addr 127.0.0.1 --master-port 29733 tests/dist_worker.py --cases /tmp/flake_cases.json > gpurun_out/r05_flake_batch.log 2>&1
echo "rc=$?"
grep -c "CASE-END" gpurun_out/r05_flake_batch.log
grep "DMP-FLAKE\|DMP-DIAG\|CASE-FAIL" -A10 gpurun_out/r05_flake_batch.log | head -80

#!/bin/bash
# round 3, GPU call A: whole GPU suite + the CG-fusion lab
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 ) > gpurun_out/r3_pytest_a.log 2>&1
tail -15 gpurun_out/r3_pytest_a.log
for n in 1000000 200000; do timeout 120 lab/cgfuse_lab $n 60; done > gpurun_out/r3_cgfuse_lab.txt 2>&1
cat gpurun_out/r3_cgfuse_lab.txt

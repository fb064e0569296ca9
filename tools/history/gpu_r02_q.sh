#!/bin/bash
# round 2, call Q: multi-rank code paths with a 1-rank RCCL group, per launch mode; plain cmdlist at world 1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r02q.log; : > $L
for m in cmdlist graph eager; do timeout 300 python tools/dist1_check.py $m 40 2>&1 | grep "^mode\|Error\|error\|Traceback" | tail -4 | cut -c1-400 >> $L; done
cat $L

#!/bin/bash
# Round 5, call O (last seconds of the budget): the bench line through the saved stdout descriptor - stdout must hold exactly one line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05o
timeout 60 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-module-path > $L.out1.txt 2> $L.err1.txt; echo "rc=$? stdout lines: $(wc -l < $L.out1.txt)"; cut -c1-160 $L.out1.txt
timeout 100 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-kernel-timer --module-steps 3 > $L.out2.txt 2> $L.err2.txt; echo "rc=$? stdout lines: $(wc -l < $L.out2.txt)"; python -c "
import json; d=json.loads(open('$L.out2.txt').readline()); print(d['ms_per_step'], {k:(v.get('ms_per_step') or v.get('error')) for k,v in d['module_path'].items() if isinstance(v,dict)})"
grep -c "RCCL version" $L.err2.txt

#!/bin/bash
# One gpurun call: GPU tests, parity report, bench line, rocprofv3 kernel stats.  Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
what="${1:-all}"
if [[ "$what" == all || "$what" == *tests* ]]; then
  timeout 900 python -m pytest tests -m gpu -q -s --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
  tail -15 $O/pytest_gpu.log
fi
if [[ "$what" == all || "$what" == *parity* ]]; then
  timeout 300 python tools/parity_report.py tiny 4 64 0.1 > $O/parity_tiny.log 2>&1; tail -30 $O/parity_tiny.log
fi
if [[ "$what" == all || "$what" == *stage* ]]; then
  timeout 300 python tools/stage_bwd_check.py tiny 4 64 > $O/stage_bwd_tiny.log 2>&1; grep -v Warning $O/stage_bwd_tiny.log | tail -40
fi
if [[ "$what" == all || "$what" == *bench* ]]; then
  timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1; echo "bench rc=$?"; tail -3 $O/bench.log
fi
if [[ "$what" == all || "$what" == *prof* ]]; then
  rm -rf $O/prof
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r01 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer > $O/prof.log 2>&1
  echo "prof rc=$?"; tail -2 $O/prof.log
  f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -40 "$f"
  find $O/prof -name '*kernel_trace.csv' -size +30M -delete
fi

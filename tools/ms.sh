#!/bin/bash
# print ms_per_step of a bench run (args passed to bench.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python bench.py --no-cpu-baseline --no-kernel-timer "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.1f samples/s  %s' % (d['ms_per_step'], d['value'], d['config'].get('launch')))"

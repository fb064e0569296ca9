#!/bin/bash
# round 6, call a: (1) the RCCL-in-graph launch mode 20 times in a row (the c10d watchdog race of round 5), (2) the option-B
# gradient exchange tests + the module tests, (3) the module path timings under the one-rank DDP wrap, (4) the whole suite, timed
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r06a; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
ok=0; bad=0
for i in $(seq 20); do
  MASTER_PORT=$((29600 + i)) timeout 180 python tools/dist1_check.py graph 4 > $O/dist1_graph_$i.log 2>&1
  rc=$?
  if [ $rc -eq 0 ] && grep -q "^mode graph -> launch graph graph_error None" $O/dist1_graph_$i.log; then ok=$((ok+1)); else bad=$((bad+1)); echo "run $i rc=$rc"; head -c 1500 $O/dist1_graph_$i.log; fi
done
echo "dist1_check graph: $ok ok, $bad bad" | tee $O/dist1_graph_summary.txt
grep -h "^mode" $O/dist1_graph_*.log | cut -c1-200 | sort | uniq -c | head -5
timeout 1500 python -m pytest tests/test_ref_loop_gpu.py tests/test_module_gpu.py -m gpu -x -q -s --timeout 900 > $O/optB_tests.log 2>&1; echo "optB tests rc=$?"
grep -E "passed|failed|error|Error|own |assert" $O/optB_tests.log | cut -c1-400 | tail -12
for opt in torch cris; do
  for se in 1 0; do
    CRIS_DDP_SELF_EXCHANGE=$se timeout 300 python bench.py --path module --ddp-one-rank --optimizer $opt --steps 20 --warmup 5 > $O/module_ddp1_${opt}_se$se.log 2>&1
    echo "module ddp1 opt=$opt self_exchange=$se rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/module_ddp1_${opt}_se$se.log | head -1)"
  done
  timeout 300 python bench.py --path module --optimizer $opt --steps 20 --warmup 5 > $O/module_bare_${opt}.log 2>&1
  echo "module bare opt=$opt rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/module_bare_${opt}.log | head -1)"
done
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 --durations=25 > $O/gpu_suite.log 2>&1; echo "suite rc=$? in $(( $(date +%s) - t0 )) s"
tail -45 $O/gpu_suite.log | cut -c1-300

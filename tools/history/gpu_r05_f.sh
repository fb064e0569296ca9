#!/bin/bash
# Round 5, call F: (1) the bench line's stdout is ONE JSON line (the one-rank DDP module runs in processes of their own);
# (2) SQ counters of the operand-path BatchNorm probe against the plain kernel on the transformed operand (why it lost)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r05f
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $L.bench_stdout.txt 2>$L.bench_stderr.txt; echo "bench rc=$? stdout lines: $(wc -l < $L.bench_stdout.txt)"; cut -c1-300 $L.bench_stdout.txt | head -3; tail -5 $L.bench_stderr.txt | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05f.bench_stdout.txt").readline())
print(d["ms_per_step"], {k:(v.get("ms_per_step") or v.get("error")) for k,v in d["module_path"].items() if isinstance(v,dict)})
PY
P=tools/probe
for sh in "8 26 256 256 3" "8 26 256 1024 1"; do
 tag=$(echo $sh | tr ' ' '_')
 for b in afuseref afuse; do
  d=gpurun_out/pmc_$b; rm -rf $d
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $d -o pmc -- $P/gemm4_probe_$b 64x64 $sh 5 > $L.pmc1_${b}_$tag.log 2>&1
  echo "== $b $sh (wave-cycle breakdown)"; python tools/pmc_sq_summary.py $(find $d -name "*_results.db" | head -1) $L.sq1_${b}_$tag.json | tail -2; rm -rf $d
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $d -o pmc -- $P/gemm4_probe_$b 64x64 $sh 5 > $L.pmc2_${b}_$tag.log 2>&1
  echo "rc=$?"; python tools/pmc_sq_summary.py $(find $d -name "*_results.db" | head -1) $L.sq2_${b}_$tag.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('$L.sq2_${b}_$tag.json'))
for k,x in d.items(): print(k[:50], {c:(round(v/ x['launches']/1e3,1) if isinstance(v,(int,float)) and c.startswith('SQ') else v) for c,v in x.items() if c.startswith('SQ') or c=='launches'})
" 2>&1 | tail -3; rm -rf $d
 done
done

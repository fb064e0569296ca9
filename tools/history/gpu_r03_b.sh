#!/bin/bash
# round 3, call B: what bounds the 8-wave GEMM loop - ablation probes (tools/probe/gemm8_probe.hip) + SQ counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; L=gpurun_out/r03b
: > $L.probe.log
for shape in "0 8 104 256 512 3" "0 8 52 512 512 3" "0 21632 1 4608 512 1" "2 8 104 512 256 3" "2 8 52 512 256 3" "1 8 104 128 128 3"; do
  for m in 0 128 1 2 4 8 16 32 64 3 11 15 144 160; do
    v=${shape%% *}
    if [ "$v" != "0" ] && [ $m != 0 ] && [ $m != 128 ] && [ $m != 1 ] && [ $m != 2 ] && [ $m != 4 ] && [ $m != 15 ]; then continue; fi
    timeout 60 tools/probe/gemm8_probe_$m $shape 20 2>&1 | grep G8PROBE >> $L.probe.log
  done
done
echo "=== probes"; cat $L.probe.log
# SQ counters of the unablated kernel on one shape (two passes: 8 SQ slots each)
cd /tmp
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -d /tmp/pmc1 -o p1 --output-format csv -- $GRAFT_REPO_ROOT/tools/probe/gemm8_probe_0 0 8 104 256 512 3 5 > /tmp/pmc1.log 2>&1
timeout 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --kernel-trace -d /tmp/pmc2 -o p2 --output-format csv -- $GRAFT_REPO_ROOT/tools/probe/gemm8_probe_0 0 8 104 256 512 3 5 > /tmp/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' > $L.pmc.log 2>&1
import csv, glob, collections
for d in ("/tmp/pmc1", "/tmp/pmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "conv_gemm8" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print("PMC %-32s mean %.4g over %d dispatches" % (k, sum(v) / len(v), len(v)))
PY
echo "=== pmc"; cat $L.pmc.log; tail -3 /tmp/pmc1.log /tmp/pmc2.log | cut -c1-300

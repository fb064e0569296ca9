"""Summarise rocprofv3 SQ counters per kernel (run ON the GPU box: the raw db exceeds gpurun's merge limit)."""
import json, sqlite3, sys
c = sqlite3.connect(sys.argv[1]).cursor()
rows = c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
d = {}
for k, cn, n, v in rows:
    key = k.split("(")[0].replace("void ", "")
    d.setdefault(key, {"launches": n})[cn] = v
json.dump(d, open(sys.argv[2], "w"), indent=1)
def g(x, cname): return x.get(cname, 0.0)
print("%-46s %6s %10s %6s %9s %7s %9s %9s" % ("kernel", "n", "wavecyc(M)", "wait%", "waitInst%", "act%", "mfmaBusy%", "ldsStall%"))
for k, x in sorted(d.items(), key=lambda kv: -g(kv[1], "SQ_WAVE_CYCLES"))[:16]:
    wc = g(x, "SQ_WAVE_CYCLES") or 1
    print("%-46s %6d %10.1f %6.1f %9.1f %7.1f %9.1f %9.1f" % (k[:46], x["launches"], wc / 1e6, 100 * g(x, "SQ_WAIT_ANY") / wc,
          100 * g(x, "SQ_WAIT_INST_ANY") / wc, 100 * g(x, "SQ_ACTIVE_INST_ANY") / wc,
          100 * g(x, "SQ_VALU_MFMA_BUSY_CYCLES") / max(g(x, "SQ_BUSY_CYCLES"), 1), 100 * g(x, "SQ_WAIT_INST_LDS") / wc))

"""Summarise rocprofv3 SQ / GRBM counters per kernel (run ON the GPU box: the raw db exceeds gpurun's merge limit).

Besides the wave-cycle breakdown this derives the MFMA utilisation AGAINST THE CHIP (north_star: "MFMA utilisation against chip
peak"): SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD with an MFMA in flight (= 32 x the number of 32x32x16 bf16 MFMAs,
MI355X_MICROARCH.md), summed over the 1024 SIMDs of the chip; the denominator is 1024 x the kernel's active cycles.  The
active cycles come from GRBM_GUI_ACTIVE - rocprofv3 sums it over the 8 XCDs, which the script checks against the kernel's wall
time from the trace (effective clock = GUI_ACTIVE / 8 / duration must land between 1.2 and 2.6 GHz; otherwise the un-divided
value is used) and reports."""
import json
import sqlite3
import sys

SIMDS = 1024
c = sqlite3.connect(sys.argv[1]).cursor()
rows = c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
d = {}
for k, cn, n, v in rows:
    key = k.split("(")[0].replace("void ", "")
    d.setdefault(key, {"launches": n})[cn] = v
try:
    for name, n, t in c.execute("select name, count(*), sum(end-start) from kernels group by name").fetchall():
        key = name.split("(")[0].replace("void ", "")
        if key in d:
            d[key]["duration_ns"] = d[key].get("duration_ns", 0) + t
            d[key]["trace_launches"] = d[key].get("trace_launches", 0) + n
except sqlite3.Error:
    pass


def g(x, cname):
    return x.get(cname, 0.0)


for k, x in d.items():
    gui, dur = g(x, "GRBM_GUI_ACTIVE"), g(x, "duration_ns")
    if gui and dur:
        clk8 = gui / 8.0 / dur                      # GHz if the counter is summed over the 8 XCDs
        per_xcd = gui / 8.0 if 1.2 <= clk8 <= 2.6 else gui
        x["effective_clock_ghz"] = per_xcd / dur
        x["gui_active_summed_over_xcds"] = bool(1.2 <= clk8 <= 2.6)
        x["mfma_util_vs_chip"] = g(x, "SQ_VALU_MFMA_BUSY_CYCLES") / (SIMDS * per_xcd)
    if dur:
        # the same ratio against wall time at the 2.4 GHz peak clock: usable for SHORT kernels too (GRBM_GUI_ACTIVE of a
        # 10-20 us dispatch is dominated by the counter-collection window: "clocks" of 20-40 GHz come out)
        x["mfma_util_vs_chip_at_2p4ghz"] = g(x, "SQ_VALU_MFMA_BUSY_CYCLES") / (SIMDS * dur * 2.4)
json.dump(d, open(sys.argv[2], "w"), indent=1)
tot_mf = sum(g(x, "SQ_VALU_MFMA_BUSY_CYCLES") for x in d.values())
tot_dur = sum(g(x, "duration_ns") for x in d.values())
if tot_dur:
    print("all kernels: MFMA busy / (1024 SIMDs x summed kernel time x 2.4 GHz) = %.1f %%" % (100 * tot_mf / (SIMDS * tot_dur * 2.4)))
print("%-40s %6s %10s %6s %9s %7s %9s %8s %9s %9s" % ("kernel", "n", "wavecyc(M)", "wait%", "waitInst%", "act%", "ldsStall%", "clk GHz", "MFMA/chip%", "@2.4GHz%"))
for k, x in sorted(d.items(), key=lambda kv: -g(kv[1], "SQ_WAVE_CYCLES"))[:18]:
    wc = g(x, "SQ_WAVE_CYCLES") or 1
    print("%-40s %6d %10.1f %6.1f %9.1f %7.1f %9.1f %8.2f %9.1f %9.1f" % (
        k[:40], x["launches"], wc / 1e6, 100 * g(x, "SQ_WAIT_ANY") / wc, 100 * g(x, "SQ_WAIT_INST_ANY") / wc,
        100 * g(x, "SQ_ACTIVE_INST_ANY") / wc, 100 * g(x, "SQ_WAIT_INST_LDS") / wc, x.get("effective_clock_ghz", 0.0),
        100 * x.get("mfma_util_vs_chip", 0.0), 100 * x.get("mfma_util_vs_chip_at_2p4ghz", 0.0)))

"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/gpu_pmc.sh).
FETCH_SIZE is in KiB and, on gfx950, reports HALF the bytes of wide coalesced reads (MI355X_MICROARCH.md, section HBM):
it is doubled here.  WRITE_SIZE (KiB) is used as reported (uncalibrated).  Writes profiles/<tag>_hbm_traffic.json:
{kernel: {launches, fetch_bytes_per_launch, write_bytes_per_launch, traffic_bytes_per_launch}}."""
import json
import sqlite3
import sys


def per_kernel(db_path, counter):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name",
                       (counter,)).fetchall()
    return {r[0]: (r[1], r[2] * 1024.0) for r in rows}


def short(name):
    return name.split("(")[0].split("<")[0].replace("void ", "")


def main(fetch_db, write_db, out, steps=0):
    """steps: training steps the traced command ran (set-up + warm-up + timed): per-step figures = totals / steps"""
    steps = int(steps)
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        nf, fb = f.get(k, (0, 0.0))
        nw, wb = w.get(k, (0, 0.0))
        fpl = 2.0 * fb / nf if nf else 0.0
        wpl = wb / nw if nw else 0.0
        key = short(k)
        if key in res:          # template variants share a short name: merge launch-weighted
            a = res[key]
            n0 = a["launches"]
            a["fetch_bytes_per_launch"] = (a["fetch_bytes_per_launch"] * n0 + fpl * nf) / (n0 + nf)
            a["write_bytes_per_launch"] = (a["write_bytes_per_launch"] * n0 + wpl * nf) / (n0 + nf)
            a["launches"] = n0 + nf
        else:
            res[key] = {"launches": nf, "fetch_bytes_per_launch": fpl, "write_bytes_per_launch": wpl}
    # the forward / input-gradient GEMM family as one entry (4-wave tiles + 8-wave tiles): what bench.py's roofline line quotes
    fam = [res[k] for k in ("conv_gemm_kernel", "conv_gemm8_kernel", "conv_gemm_group_kernel", "conv_gemm8_group_kernel") if k in res]
    if fam:
        n = sum(a["launches"] for a in fam)
        res["conv_gemm (all tile kernels)"] = {
            "launches": n,
            "fetch_bytes_per_launch": sum(a["fetch_bytes_per_launch"] * a["launches"] for a in fam) / n,
            "write_bytes_per_launch": sum(a["write_bytes_per_launch"] * a["launches"] for a in fam) / n}
    for a in res.values():
        a["traffic_bytes_per_launch"] = a["fetch_bytes_per_launch"] + a["write_bytes_per_launch"]
    total = sum(a["traffic_bytes_per_launch"] * a["launches"] for k, a in res.items() if k != "conv_gemm (all tile kernels)")
    import os
    stamp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".source_commit")
    commit = open(stamp).read().strip() if os.path.exists(stamp) else None      # (written by tools/gpurun.sh before the snapshot)
    json.dump({"commit": commit, "note": "FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE as reported; eager launches, %s" % fetch_db,
               "steps": steps, "total_bytes_per_step": total / steps if steps else None, "kernels": res}, open(out, "w"), indent=1)
    if steps:
        print("whole step: %.2f GB memory-side traffic per step (%d steps traced)" % (total / steps / 1e9, steps))
    for k, a in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print("%-28s n=%5d fetch/launch %8.2f MB write/launch %8.2f MB" % (k, a["launches"], a["fetch_bytes_per_launch"] / 1e6,
                                                                             a["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:5])

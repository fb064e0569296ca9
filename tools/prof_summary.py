"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd .db) into a CSV like rocprofv3's kernel_stats.csv.
usage: python tools/prof_summary.py gpurun_out/prof/r01_results.db profiles/r01_kernel_stats.csv [steps_in_run]"""
import sqlite3
import sys


def main(db_path, out_path, steps=None):
    c = sqlite3.connect(db_path).cursor()
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                     "group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out_path, "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs%s\n" % (",CallsPerStep,MsPerStep" if steps else ""))
        for name, n, t, avg, mn, mx in rows:
            extra = ",%.1f,%.4f" % (n / steps, t / 1e6 / steps) if steps else ""
            f.write('"%s",%d,%d,%.1f,%.2f,%d,%d%s\n' % (name, n, t, avg, 100.0 * t / tot, mn, mx, extra))
    print("kernels %d, launches %d, total kernel time %.2f ms%s" % (
        len(rows), sum(r[1] for r in rows), tot / 1e6, (", %.3f ms/step" % (tot / 1e6 / steps)) if steps else ""))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else None)

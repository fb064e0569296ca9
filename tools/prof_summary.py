"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd .db) into a CSV like rocprofv3's kernel_stats.csv.
usage: python tools/prof_summary.py gpurun_out/prof/r01_results.db profiles/r01_kernel_stats.csv [steps [marker]]

Without `marker` every launch of the process is counted and divided by `steps` (so set-up work - the zero fills of the
optimiser state, the parameter copies - is smeared over the steps).  With `marker` (a kernel-name prefix that occurs
exactly once per step, e.g. "void adam_kernel<1>") only the last `steps` steps are counted: the launches that start after
the end of marker launch number (last - steps) and up to the end of the last one - the steady state the bench times."""
import sqlite3
import sys


def main(db_path, out_path, steps=None, marker=None):
    c = sqlite3.connect(db_path).cursor()
    where = ""
    if marker:
        marks = c.execute("select end from kernels where name like ? order by start", (marker + "%",)).fetchall()
        steps = int(steps)
        if len(marks) <= steps:
            raise SystemExit("only %d launches of %r in the trace, need more than %d" % (len(marks), marker, steps))
        where = "where start > %d and start <= %d" % (marks[-steps - 1][0], marks[-1][0])
    # how much of the window has at least one kernel running (union of the intervals), and how much none
    iv = c.execute("select start, end from kernels %s order by start" % where).fetchall()
    busy, cur_s, cur_e = 0, None, None
    for s_, e_ in iv:
        if cur_e is None or s_ > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    if cur_e is not None:
        busy += cur_e - cur_s
    span = (iv[-1][1] - iv[0][0]) if iv else 0
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels %s "
                     "group by name order by 3 desc" % where).fetchall()
    tot = sum(r[2] for r in rows)
    with open(out_path, "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs%s\n" % (",CallsPerStep,MsPerStep" if steps else ""))
        for name, n, t, avg, mn, mx in rows:
            extra = ",%.1f,%.4f" % (n / steps, t / 1e6 / steps) if steps else ""
            f.write('"%s",%d,%d,%.1f,%.2f,%d,%d%s\n' % (name, n, t, avg, 100.0 * t / tot, mn, mx, extra))
    print("kernels %d, launches %d%s, total kernel time %.2f ms%s" % (
        len(rows), sum(r[1] for r in rows), (" (%.1f per step)" % (sum(r[1] for r in rows) / steps)) if steps else "", tot / 1e6,
        (", %.3f ms/step" % (tot / 1e6 / steps)) if steps else ""))
    if steps and span:
        print("window %.3f ms/step: some kernel running %.3f ms/step, none %.3f ms/step (launch gaps), overlapped (two or more) %.3f ms/step"
              % (span / 1e6 / steps, busy / 1e6 / steps, (span - busy) / 1e6 / steps, (tot - busy) / 1e6 / steps))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else None, sys.argv[4] if len(sys.argv) > 4 else None)

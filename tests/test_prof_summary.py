"""tools/prof_summary.py: the steady-state window (launches between marker launches) and the busy / idle accounting, on a
synthetic rocpd-style database."""
import csv
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import prof_summary  # noqa: E402


def _db(path, rows):
    con = sqlite3.connect(path)
    con.execute("create table kernels (name text, start integer, end integer)")
    con.executemany("insert into kernels values (?, ?, ?)", rows)
    con.commit()
    con.close()


def test_window_counts_whole_steps_only(tmp_path, capsys):
    U = 1000000                                                                        # 1 ms in ns
    rows = [("fill", 0, 10 * U), ("fill", 20 * U, 30 * U), ("fill", 40 * U, 50 * U)]   # set-up launches
    t = 100 * U
    for step in range(5):                                                              # 5 steps: a, a, b (overlapping a), marker
        rows += [("a", t, t + 10 * U), ("a", t + 20 * U, t + 30 * U), ("b", t + 25 * U, t + 45 * U),
                 ("void marker<1>(int)", t + 50 * U, t + 60 * U)]
        t += 100 * U
    db, out = str(tmp_path / "k.db"), str(tmp_path / "k.csv")
    _db(db, rows)
    prof_summary.main(db, out, 3, "void marker<1>")
    got = {r["Name"]: r for r in csv.DictReader(open(out))}
    assert "fill" not in got                                                           # set-up is outside the window
    assert float(got["a"]["CallsPerStep"]) == 2.0 and float(got["b"]["CallsPerStep"]) == 1.0
    assert int(got["a"]["Calls"]) == 6 and int(got["void marker<1>(int)"]["Calls"]) == 3
    text = capsys.readouterr().out
    assert "launches 12 (4.0 per step)" in text
    # per step: a 10 + (a|b union 20..45 = 25) + marker 10 = 45 ms busy; kernel time 50 ms, so 5 ms of it overlapped
    assert "some kernel running 45.000 ms/step" in text and "overlapped (two or more) 5.000 ms/step" in text
    assert abs(float(got["b"]["MsPerStep"]) - 20.0) < 1e-9


def test_without_marker_everything_is_counted(tmp_path):
    db, out = str(tmp_path / "k.db"), str(tmp_path / "k.csv")
    U = 1000000
    _db(db, [("x", 0, 5 * U), ("x", 10 * U, 15 * U), ("y", 20 * U, 30 * U)])
    prof_summary.main(db, out, 2.0)
    got = {r["Name"]: r for r in csv.DictReader(open(out))}
    assert float(got["x"]["CallsPerStep"]) == 1.0 and float(got["y"]["MsPerStep"]) == 5.0

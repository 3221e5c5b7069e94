#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace results database -> CSV.
usage: python tools/rocprof_db_stats.py <results.db> [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from {kt} d join {ks} s "
                       "on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    out = ["name,calls,total_ns,avg_ns,percent"]
    for n, c, t, a in rows:
        out.append('"%s",%d,%d,%.0f,%.2f' % (n[:160], c, t, a, 100 * t / tot))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print("\n".join(l[:200] for l in out[:30]))
    print("total ms", tot / 1e6)


if __name__ == "__main__":
    main()

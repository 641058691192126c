#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd SQLite database (same columns as --stats)."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
        "from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (disp, sym)).fetchall()
    total = float(sum(r[2] for r in rows)) or 1.0
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for name, calls, tot, avg, mn, mx in rows:
        lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (name, calls, tot, avg, 100.0 * tot / total, mn, mx))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

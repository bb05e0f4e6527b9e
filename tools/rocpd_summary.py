#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (``*_results.db``) into a per-kernel stats table
(name, calls, total ms, average ms, percent) - the same content as ``rocprofv3 --stats``'s kernel_stats.csv."""
import sqlite3
import sys


def main(db_path, out_path=None, limit=90):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["# source: %s (rocprofv3 --kernel-trace --stats, rocpd top_kernels view; durations in ms)" % db_path,
             "calls,total_ms,avg_ms,percent,kernel"]
    for name, calls, total, avg, pct in rows[:limit]:
        name = name if len(name) < 140 else name[:137] + "..."
        lines.append("%d,%.3f,%.3f,%.2f,\"%s\"" % (calls, total / 1e3, avg / 1e3, pct, name))
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])

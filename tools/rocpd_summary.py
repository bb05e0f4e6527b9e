#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (``*_results.db``) into a per-kernel stats table
(name, calls, total ms, average ms, percent) - the same content as ``rocprofv3 --stats``'s kernel_stats.csv.

    rocpd_summary.py <db> [out.csv] [--skip-steps N] [--marker NAME]

``--skip-steps N``: WARM-ONLY statistics - dispatches before the (N+1)-th launch of the once-per-step marker kernel (default
``scene_pairs_kernel``: ``flatten_scene``, the first launch of every training / evaluation step - note that ``bench.py`` also flattens once before its loop) are dropped, so that the cold first call of
every kernel (code-object load, cold caches, workspace zeroing) does not sit in the averages; an extra column gives ms per step.
"""
import sqlite3
import sys


def main(argv):
    skip, marker = 0, "scene_pairs_kernel"
    pos = []
    it = iter(argv)
    for a in it:
        if a == "--skip-steps":
            skip = int(next(it))
        elif a == "--marker":
            marker = next(it)
        else:
            pos.append(a)
    db_path, out_path, limit = pos[0], (pos[1] if len(pos) > 1 else None), 90
    cur = sqlite3.connect(db_path).cursor()
    head = "# source: %s (rocprofv3 --kernel-trace --stats, rocpd %s; durations in ms)"
    if skip <= 0:
        rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        lines = [head % (db_path, "top_kernels view"), "calls,total_ms,avg_ms,percent,kernel"]
        for name, calls, total, avg, pct in rows[:limit]:
            name = name if len(name) < 140 else name[:137] + "..."
            lines.append("%d,%.3f,%.3f,%.2f,\"%s\"" % (calls, total / 1e3, avg / 1e3, pct, name))
    else:
        marks = [r[0] for r in cur.execute("select start from kernels where name like ? order by start", ("%" + marker + "%",))]
        if len(marks) <= skip:
            raise SystemExit("only %d launches of the marker kernel %s: cannot skip %d steps" % (len(marks), marker, skip))
        t0, steps = marks[skip], len(marks) - skip
        rows = list(cur.execute("select name, count(*), sum(end - start) from kernels where start >= ? group by name "
                                "order by sum(end - start) desc", (t0,)))
        grand = float(sum(r[2] for r in rows)) or 1.0
        lines = [head % (db_path, "kernels view, WARM ONLY: the last %d of %d steps" % (steps, len(marks))),
                 "calls,total_ms,avg_ms,percent,ms_per_step,kernel"]
        for name, calls, total in rows[:limit]:
            name = name if len(name) < 140 else name[:137] + "..."
            lines.append("%d,%.3f,%.3f,%.2f,%.3f,\"%s\"" % (calls, total / 1e6, total / 1e6 / calls, 100.0 * total / grand, total / 1e6 / steps, name))
        lines.append("# all kernels: %.3f ms per step" % (grand / 1e6 / steps))
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1:])

#!/usr/bin/env python3
"""Timeline of ONE warm step from a rocprofv3 rocpd database: every kernel dispatch between two consecutive launches of the step
marker (``scene_pairs_kernel`` = ``flatten_scene``), with start / duration relative to the step's first dispatch, the HIP stream
and how much of the dispatch overlaps another one.  Shows what the two-stream backward really overlaps and where the GPU idles.

    rocpd_timeline.py <db> [out.csv] [--step N] [--marker NAME]     (N counted over the marker launches, default: the last complete step)
"""
import sqlite3
import sys


def main(argv):
    step, pos, marker = None, [], "scene_pairs_kernel"
    it = iter(argv)
    for a in it:
        if a == "--step":
            step = int(next(it))
        elif a == "--marker":
            marker = next(it)
        else:
            pos.append(a)
    db, out = pos[0], (pos[1] if len(pos) > 1 else None)
    cur = sqlite3.connect(db).cursor()
    marks = [r[0] for r in cur.execute("select start from kernels where name like ? order by start", ("%" + marker + "%",))]
    if len(marks) < 2:
        raise SystemExit("need at least two step markers")
    k = len(marks) - 2 if step is None else step
    t0, t1 = marks[k], marks[k + 1]
    rows = list(cur.execute("select name, start, end, stream_id from kernels where start >= ? and start < ? order by start", (t0, t1)))
    ev = sorted([(s, 1) for _, s, e, _ in rows] + [(e, -1) for _, s, e, _ in rows])
    # time covered by >= 1 and by >= 2 dispatches
    busy = both = 0
    depth, last = 0, ev[0][0]
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            both += t - last
        depth += d
        last = t
    span = max(e for _, _, e, _ in rows) - t0
    lines = ["# %s step %d of %d: %d dispatches, span %.3f ms, GPU busy %.3f ms (idle %.3f), two or more dispatches in flight %.3f ms, "
             "sum of durations %.3f ms" % (db, k, len(marks) - 1, len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6, both / 1e6,
                                           sum(e - s for _, s, e, _ in rows) / 1e6),
             "start_ms,duration_ms,stream,kernel"]
    for name, s, e, st in rows:
        name = name if len(name) < 100 else name[:97] + "..."
        lines.append("%.4f,%.4f,%s,\"%s\"" % ((s - t0) / 1e6, (e - s) / 1e6, st, name))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1:])

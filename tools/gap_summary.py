"""Per-kernel duration AND the idle time around each launch, from a rocprofv3 --kernel-trace database (rocpd .db):
   python tools/gap_summary.py <results.db> <out.csv> [skip_first_n]
For every kernel name: launches, mean duration, mean gap between the previous kernel's end and this kernel's start (the
drain + dispatch a dependent launch pays), and the sum of both -- what a launch costs a single in-order stream."""
import csv
import sqlite3
import sys


def main(db, out, skip=0):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()[skip:]
    agg = {}
    prev_end = None
    span0, span1 = rows[0][1], rows[-1][2]
    for name, s, e in rows:
        gap = max(0, s - prev_end) if prev_end is not None else 0
        a = agg.setdefault(name, [0, 0, 0])
        a[0] += 1; a[1] += e - s; a[2] += gap
        prev_end = max(prev_end or e, e)
    tot = span1 - span0
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "AvgDurationNs", "AvgGapBeforeNs", "TotalDurationPct", "TotalGapPct"])
        for name, (n, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
            w.writerow([name[:110], n, "%.0f" % (d / n), "%.0f" % (g / n), "%.3f" % (100.0 * d / tot), "%.3f" % (100.0 * g / tot)])
        w.writerow(["TOTAL span ns", len(rows), tot, "", "%.3f" % (100.0 * sum(a[1] for a in agg.values()) / tot),
                    "%.3f" % (100.0 * sum(a[2] for a in agg.values()) / tot)])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)

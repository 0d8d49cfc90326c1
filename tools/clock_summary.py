"""Effective clock and matrix-pipe occupancy per kernel from ONE rocprofv3 --pmc pass (GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES,
--kernel-trace) over a reduced step:   python tools/clock_summary.py <results.db> <out.csv>
  clock    = sum GRBM_GUI_ACTIVE / 8 / sum duration        (the counter is summed over the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back;
                                                            reads high on dispatches shorter than ~0.3 ms, and LOW by the
                                                            10-16 % a profiled dispatch takes longer than an un-profiled one:
                                                            not the clock the kernel runs at -- tools/clock_probe.py measures that)
  mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum GRBM_GUI_ACTIVE / 8)"""
import csv
import sqlite3
import sys


def main(db, out):
    cur = sqlite3.connect(db).cursor()
    dur = {r[0]: (r[1], r[2]) for r in cur.execute("select name, count(*), sum(duration) from kernels group by name")}
    cnt = {}
    for name, counter, total in cur.execute("select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"):
        cnt.setdefault(name, {})[counter] = total
    tot = sum(v[1] for v in dur.values())
    rows = []
    for name, (calls, ns) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        c = cnt.get(name, {})
        g, m = c.get("GRBM_GUI_ACTIVE"), c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if not g or 100.0 * ns / tot < 0.3:
            continue
        cyc = g / 8.0
        rows.append([name, calls, "%.1f" % (ns / calls / 1e3), "%.2f" % (100.0 * ns / tot), "%.3f" % (cyc / ns), "%.3f" % ((m or 0.0) / (1024.0 * cyc))])
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Dispatches", "AverageUs", "SharePct", "EffectiveClockGHz", "MfmaBusyFraction"])
        w.writerows(rows)
    for r in rows:
        print("%-100s %6s x %9s us  %5s %%  clock %s GHz  mfma busy %s" % (r[0][:100], r[1], r[2], r[3], r[4], r[5]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

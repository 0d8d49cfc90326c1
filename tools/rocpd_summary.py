"""Summarises rocprofv3 output databases (rocpd .db, the ROCm 7.2 default format) into the small CSVs that
are committed under profiles/:   python tools/rocpd_summary.py <results.db> <out_prefix>
  <out_prefix>_kernel_stats.csv   per-kernel calls / total / average / share (what --stats prints)
  <out_prefix>_by_grid.csv        per (kernel, grid) calls / average: one row per UNet layer shape
  <out_prefix>_counters.csv       per (kernel, counter) average value (only for --pmc runs)"""
import csv
import sqlite3
import sys


def main(db_path, prefix):
    cur = sqlite3.connect(db_path).cursor()
    tot = cur.execute("select sum(duration) from kernels").fetchone()[0] or 1
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                             "group by name order by sum(duration) desc"):
            w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.4f" % (100.0 * r[2] / tot), r[4], r[5]])
    with open(prefix + "_by_grid.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "GridX", "GridY", "GridZ", "WorkgroupX", "VGPR", "LDS", "Calls", "AverageNs", "Percentage"])
        for r in cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, lds_size, count(*), avg(duration), "
                             "sum(duration) from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc"):
            if 100.0 * r[9] / tot >= 0.05:
                w.writerow([r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], "%.1f" % r[8], "%.3f" % (100.0 * r[9] / tot)])
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    if rows:
        with open(prefix + "_counters.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel", "Counter", "AverageValuePerDispatch", "Dispatches"])
            for r in rows:
                w.writerow([r[0], r[1], "%.6g" % r[2], r[3]])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

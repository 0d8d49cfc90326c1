"""HBM traffic of the dominant kernel from the two whole-step counter passes of tools/profile_round.sh
(rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE over a reduced `bench.py` step with the headline's mix of forwards, separate runs):

    python tools/traffic_summary.py <fetch results.db> <write results.db> <tag>   ->  profiles/<tag>_traffic.json
                                                                                       profiles/<tag>_hbm_by_kernel.csv

Counters are in KB.  FETCH_SIZE is corrected x2: on gfx950 it reports exactly half of the bytes of the load widths this
kernel uses (MI355X_MICROARCH.md, HBM section; re-checked on this pool with tools/ubench/fetch_calib.hip in round 1:
profiles/r01l_fetch_calib_counters.csv); WRITE_SIZE reads bytes exactly for 16-byte-per-lane stores.  The summary records
the sha256 of csrc/conv_ws.hip and the mode: bench.py reports `roofline.traffic` only from a summary taken on the very
kernel source and mode it runs."""
import csv
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, avg, n in cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? "
                                    "group by kernel_name", (counter,)):
        out[name] = (avg * 1024.0, n)
    dur = {}
    for name, tot, n in cur.execute("select name, sum(duration), count(*) from kernels group by name"):
        dur[name] = (tot, n)
    return out, dur


def main(fetch_db, write_db, tag):
    f, dur = per_kernel(fetch_db, "FETCH_SIZE")
    w, _ = per_kernel(write_db, "WRITE_SIZE")
    dom = [k for k in f if "conv_wino2_kernel" in k] or [k for k in f if "conv_wino_kernel" in k]
    if not dom or os.environ.get("IPDM_CONV_NO_WINO"):
        dom = [k for k in f if "conv_ws_kernel<3, 1," in k]
    nl = sum(f[k][1] for k in dom)
    fetch = sum(f[k][0] * f[k][1] for k in dom) / nl
    write = sum(w[k][0] * w[k][1] for k in dom if k in w) / nl
    src = "conv_wino2.hip" if any("conv_wino2_kernel" in k for k in dom) else ("conv_wino.hip" if any("conv_wino_kernel" in k for k in dom) else "conv_ws.hip")
    with open(os.path.join(ROOT, "ipdm-pytorch_amd", "csrc", src), "rb") as fh:
        sha = hashlib.sha256(fh.read()).hexdigest()[:16]
    mode = "exact-f32"
    if os.environ.get("IPDM_CONV_NO_WINO"):
        mode += "-nowino"
    d = {"tag": tag, "kernel": {"conv_wino2.hip": "conv_wino2_kernel (wide 3x3 stride-1, Winograd domain, 128-cout tiles)",
                                "conv_wino.hip": "conv_wino_kernel (wide 3x3 stride-1, Winograd domain, 64-cout tiles)"}.get(src, "conv_ws_kernel<3,1,*> (3x3 stride-1)"),
         "mode": mode, "kernel_source_sha16": sha,
         "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over python3 bench.py --steps 1 --warmup 0 "
                   "--t_start_proj 3 --t_start_img 2 --no-ultra (B=8: the headline's 3:2 mix of proj and img UNet forwards)",
         "launches": nl, "fetch_size_bytes_per_launch_raw": fetch, "fetch_calibration_factor": 2.0,
         "fetch_bytes_per_launch": 2.0 * fetch, "write_bytes_per_launch": write,
         "traffic_bytes_per_launch": 2.0 * fetch + write}
    with open(os.path.join(ROOT, "profiles", tag + "_traffic.json"), "w") as fh:
        json.dump(d, fh, indent=1)
    with open(os.path.join(ROOT, "profiles", tag + "_hbm_by_kernel.csv"), "w", newline="") as fh:
        cw = csv.writer(fh)
        cw.writerow(["Kernel", "Launches", "TotalMs(counter pass)", "FETCH_SIZE_GB(raw)", "WRITE_SIZE_GB", "TB/s(2*fetch+write)/time"])
        rows = []
        for k in f:
            tot_ns, n = dur.get(k, (0, 0))
            fg = f[k][0] * f[k][1] / 1e9
            wg = w.get(k, (0, 0))[0] * w.get(k, (0, 0))[1] / 1e9
            rows.append((tot_ns, k, n, fg, wg))
        for tot_ns, k, n, fg, wg in sorted(rows, reverse=True)[:40]:
            cw.writerow([k[:90], n, "%.1f" % (tot_ns / 1e6), "%.1f" % fg, "%.1f" % wg,
                         "%.2f" % ((2 * fg + wg) / max(tot_ns / 1e9, 1e-9) / 1e3)])
    print(json.dumps(d))


if __name__ == "__main__":
    main(*sys.argv[1:4])

#!/usr/bin/env python3
"""Turns the rocprofv3 (rocpd sqlite) outputs of tools/profile_round.sh into the small files committed under profiles/.

  tools/summarize_profile.py gpurun_out/<tag> profiles/<tag>

Writes
  <prefix>_kernel_stats.csv      per-kernel calls / total / average / share  (the `--stats` summary of the bench command)
  <prefix>_kernel_stats_alone.csv  the same averages over the launches that ran with no other kernel in flight (two-stream backward)
  <prefix>_hbm_traffic.csv       per-kernel FETCH_SIZE and WRITE_SIZE per launch (separate --pmc passes), with the
                                 gfx950 correction of MI355X_MICROARCH.md "HBM" (FETCH_SIZE x 2) and the two
                                 kernels of the same run whose byte counts are known exactly as calibration rows
  <prefix>_traffic.json          what bench.py reports as roofline.traffic (dominant kernel, bytes per launch)
  <prefix>_sq_<probe>.csv        issue / stall counters of a single-kernel probe, if that pass exists
"""
import csv
import glob
import json
import os
import sqlite3
import sys

DOMINANT = "conv_dma_kernel<32>"   # prefix of the dominant kernel symbol (bench.py DOMINANT_SYMBOL)
GRAD_FLOATS = 44570341          # flat parameter / gradient buffer of the benchmark config


def db(path):
    f = glob.glob(os.path.join(path, "*_results.db"))
    return sqlite3.connect(f[0]) if f else None


def short(name):
    name = name.replace("void ", "")
    depth = 0
    for i, ch in enumerate(name):      # cut the argument list, keep template arguments
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def kernel_stats(con, out):
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
                       "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, c, t, a, mn, mx in rows:
            w.writerow([short(n), c, int(t), round(a, 1), round(100.0 * t / total, 4), mn, mx])
    return {short(n): (c, a) for n, c, t, a, mn, mx in rows}


def alone_stats(con, out):
    """Per kernel: launches that ran with NO other kernel in flight (by start / end time stamps) -- the backward's two streams put a
    neighbour beside many launches, and a launch that shares the chip reads longer; the roofline of a symbol is the time of its
    launches that ran alone (bench.py measures the dominant symbol on those too)."""
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    import bisect
    starts = [r[1] for r in rows]
    # running maximum of the end times of earlier launches
    agg = {}
    max_end = 0
    prev_max = []
    for n, s, e in rows:
        prev_max.append(max_end)
        max_end = max(max_end, e)
    for i, (n, s, e) in enumerate(rows):
        overl = prev_max[i] > s                                   # an earlier launch still running
        j = bisect.bisect_right(starts, s, lo=i + 1)              # launches starting at the same time stamp
        k = bisect.bisect_left(starts, e, lo=i + 1)               # later launches that start before this one ends
        overl = overl or k > i + 1 or j > i + 1
        a = agg.setdefault(short(n), [0, 0, 0, 0])
        a[0] += 1
        a[1] += e - s
        if not overl:
            a[2] += 1
            a[3] += e - s
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "AverageNs", "CallsAlone", "AverageNsAlone"])
        for n, (c, t, ca, ta) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([n, c, round(t / c, 1), ca, round(ta / ca, 1) if ca else ""])
    return agg


def counter_avgs(con, counter):
    q = ("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name")
    return {short(n): (c, v) for n, c, v in con.execute(q, (counter,))}


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    prof = sys.argv[sys.argv.index("--prof") + 1] if "--prof" in sys.argv else "prof"      # kernel-trace pass directory name
    stats = {}
    con = db(os.path.join(src, prof))
    if con:
        stats = kernel_stats(con, prefix + "_kernel_stats.csv")
        alone = alone_stats(con, prefix + "_kernel_stats_alone.csv")
        for n, (c, t, ca, ta) in alone.items():
            if n.startswith(DOMINANT):
                print("%s: %d launches, average %.1f us; %d ran alone, average %.1f us" % (n, c, t / c / 1e3, ca, ta / max(ca, 1) / 1e3))
    cf, cw = db(os.path.join(src, "pmc_fetch")), db(os.path.join(src, "pmc_write"))
    if cf and cw:
        fetch, write = counter_avgs(cf, "FETCH_SIZE"), counter_avgs(cw, "WRITE_SIZE")
        known = {   # kernels of the same run whose traffic is known exactly (calibration of the counters)
            "sumsq_stage1": (GRAD_FLOATS * 4, 0, "reads the flat gradient once, 16 B/lane"),
            "sgd_apply_kernel": (2 * GRAD_FLOATS * 4, GRAD_FLOATS * 4, "reads w and g, writes w, 4 B/lane"),
        }
        with open(prefix + "_hbm_traffic.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Launches", "FETCH_SIZE_KB_raw", "WRITE_SIZE_KB_raw", "read_bytes_corrected(x2)", "write_bytes",
                        "known_read_bytes", "known_write_bytes", "note"])
            for n in sorted(fetch, key=lambda k: -fetch[k][1] * fetch[k][0]):
                c, fk = fetch[n]
                wk = write.get(n, (0, 0.0))[1]
                kr, kw, note = known.get(n, ("", "", ""))
                w.writerow([n, c, round(fk, 1), round(wk, 1), int(fk * 1024 * 2), int(wk * 1024), kr, kw, note])
        dom = [n for n in fetch if n.startswith(DOMINANT)]
        if dom:
            DOM = dom[0]
            fk, wk = fetch[DOM][1], write.get(DOM, (0, 0.0))[1]
            cal_r = fetch.get("sumsq_stage1", (0, 0))[1] * 1024 * 2 / (GRAD_FLOATS * 4)
            cal_w = write.get("sgd_apply_kernel", (0, 0))[1] * 1024 / (GRAD_FLOATS * 4)
            rec = {"kernel": DOM, "read_bytes_per_launch": int(fk * 1024 * 2), "write_bytes_per_launch": int(wk * 1024),
                   "bytes_per_launch": int(fk * 1024 * 2 + wk * 1024), "launches_averaged": fetch[DOM][0],
                   "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 2 --warmup 1`; "
                             "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B)",
                   "calibration": {"sumsq_stage1 corrected read / known": round(cal_r, 3),
                                   "sgd_apply write / known": round(cal_w, 3)}}
            with open(prefix + "_traffic.json", "w") as f:
                json.dump(rec, f, indent=1)
            print(json.dumps(rec))
    for d in glob.glob(os.path.join(src, "pmc_sq_*")):
        if not os.path.isdir(d):
            continue
        con = db(d)
        if not con:
            continue
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name")
        with open(prefix + "_" + os.path.basename(d)[4:] + ".csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Counter", "Launches", "AvgValue"])
            for n, cn, c, v in con.execute(q):
                if "mfma" in n or "wgrad" in n or "pool" in n or "lrn" in n or "conv_" in n:
                    w.writerow([short(n), cn, c, round(v, 1)])


if __name__ == "__main__":
    main()

"""Timeline of one train step from a rocprofv3 --kernel-trace CSV: per launch its queue, start offset, duration, the idle gap on its
own queue before it, and how much of it ran beside a launch of the other queue.  Steps are delimited by sgd_apply_kernel.

    python tools/timeline.py gpurun_out/<tag>/trace_c8/c8_kernel_trace.csv [--step -2] [--min-us 0]
"""
import argparse
import csv
import re


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--step", type=int, default=-2, help="which train step (index into the sgd_apply-delimited list)")
    ap.add_argument("--min-us", type=float, default=0.0)
    a = ap.parse_args()
    if a.csv.endswith(".csv"):
        rows = [r for r in csv.DictReader(open(a.csv)) if r["Kind"] == "KERNEL_DISPATCH"]
    else:                                   # a rocprofv3 results database (or the directory that holds one)
        import glob, os, sqlite3
        db = a.csv if a.csv.endswith(".db") else sorted(glob.glob(os.path.join(a.csv, "**", "*_results.db"), recursive=True))[-1]
        rows = [{"Kernel_Name": n, "Start_Timestamp": s, "End_Timestamp": e, "Queue_Id": str(q), "grid": "%dx%dx%d" % (gx // max(wx, 1), gy, gz)}
                for n, s, e, q, gx, gy, gz, wx in sqlite3.connect(db).execute("select name, start, end, queue_id, grid_x, grid_y, grid_z, workgroup_x from kernels")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "sgd_apply_kernel" in r["Kernel_Name"] or "adam_apply" in r["Kernel_Name"]]
    if len(ends) < 3:
        raise SystemExit("fewer than 3 optimizer launches in the trace")
    hi = ends[a.step]
    lo = ends[a.step - 1] + 1
    step = rows[lo:hi + 1]
    t0 = int(step[0]["Start_Timestamp"])
    last_end = {}
    busy = []
    print("%-64s %2s %9s %8s %8s %8s" % ("kernel", "q", "start us", "dur us", "gap us", "beside"))
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = r["Queue_Id"]
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        ov = 0
        for o in step:
            if o["Queue_Id"] != q:
                ov += max(0, min(e, int(o["End_Timestamp"])) - max(s, int(o["Start_Timestamp"])))
        busy.append((s, e))
        if (e - s) / 1e3 >= a.min_us:
            print("%-64s %2s %9.1f %8.1f %8.1f %7.0f%%  %s" % (short(r["Kernel_Name"]), q[-2:], (s - t0) / 1e3, (e - s) / 1e3, gap,
                                                              100.0 * ov / max(1, e - s), r.get("grid", "")))
    busy.sort()
    union, cur_s, cur_e = 0, busy[0][0], busy[0][1]
    for s, e in busy[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    span = max(e for _, e in busy) - t0
    print("step span %.1f us, some kernel running %.1f us (idle %.1f us), sum of durations %.1f us, %d launches"
          % (span / 1e3, union / 1e3, (span - union) / 1e3, sum(e - s for s, e in busy) / 1e3, len(step)))


if __name__ == "__main__":
    main()

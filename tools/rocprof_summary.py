#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2) rocpd SQLite output into the text kept under profiles/.
Usage: rocprof_summary.py <results.db> [more.db ...]   (kernel stats; PMC counters if present)"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print(f"== {path}")
        print("-- kernel stats (top_kernels): name | calls | total_ms | avg_us | %")
        for name, calls, total, avg, pct in cur.execute(
                "select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
            # the top_kernels view reports microseconds
            print(f"{name[:110]:110s} | {calls:5d} | {total/1e3:10.3f} | {avg:10.2f} | {pct:6.2f}")
        # per-dispatch durations: a sort also enqueues launches that exit on their flag word (the plan the device did not choose) —
        # `working` = dispatches above 5 % of the kernel's longest one; their mean is what bench.py's HIP events measure
        try:
            kc = [d[0] for d in cur.execute("select * from kernels limit 1").description]
            nm = "name" if "name" in kc else "kernel_name"
            if "duration" in kc:
                q = f"select {nm}, duration from kernels"
            else:
                q = f"select {nm}, (end - start) from kernels"
            per = {}
            for name, dur in cur.execute(q):
                per.setdefault(name[:110], []).append(float(dur))
            print("-- per-dispatch durations (view `kernels`, ns): name | dispatches | mean_us | working dispatches | mean_us over the working ones")
            for name, ds in sorted(per.items(), key=lambda kv: -sum(kv[1])):
                work = [x for x in ds if x > 0.05 * max(ds)] or ds
                print(f"{name:110s} | {len(ds):5d} | {sum(ds)/len(ds)/1e3:10.2f} | {len(work):5d} | {sum(work)/len(work)/1e3:10.2f}")
        except Exception as e:  # noqa: BLE001
            print(f"-- per-dispatch durations unavailable: {e}")
        try:
            cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
            rows = list(cur.execute("select * from counters_collection"))
        except Exception:
            rows = []
        if rows:
            ik = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name")
            ic = cols.index("counter_name")
            iv = cols.index("value") if "value" in cols else cols.index("counter_value")
            agg = {}
            for r in rows:
                key = (r[ik][:110], r[ic])
                agg.setdefault(key, []).append(float(r[iv]))
            # `working` = dispatches above 5 % of the kernel's largest value: a sort enqueues launches that exit on their flag word
            # (the plan the device did not choose); they would dilute a plain mean
            print("-- PMC counters: kernel | counter | dispatches | mean value per dispatch | working dispatches | mean over the working ones")
            for (k, c), vals in sorted(agg.items()):
                work = [v for v in vals if v > 0.05 * max(vals)] or vals
                print(f"{k:110s} | {c} | {len(vals)} | {sum(vals)/len(vals):.1f} | {len(work)} | {sum(work)/len(work):.1f}")


if __name__ == "__main__":
    main()

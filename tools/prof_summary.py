#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd SQLite output) into a small text file for profiles/.

  python tools/prof_summary.py gpurun_out/prof_x/bench_results.db profiles/r01_kernel_trace.txt [--pmc]
Kernel-trace runs: per-kernel calls / total / average duration (the `top_kernels` view).
PMC runs (--pmc): per-kernel mean of every collected counter per dispatch.
"""
import sqlite3
import sys


def short(name):
    name = name.replace("cs::", "")
    return name if len(name) < 90 else name[:87] + "..."


def main():
    db, out = sys.argv[1], sys.argv[2]
    pmc = "--pmc" in sys.argv
    c = sqlite3.connect(db)
    lines = []
    if not pmc:
        rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
        lines.append(f"# rocprofv3 --kernel-trace --stats summary of {db} (durations in us)")
        lines.append(f"{'kernel':90s} {'calls':>6s} {'total_us':>14s} {'avg_us':>14s} {'pct':>7s}")
        other_calls = other_tot = 0
        for name, calls, tot, avg, pct in rows:
            if "cs::" in name:
                lines.append(f"{short(name):90s} {calls:6d} {tot:14.0f} {avg:14.1f} {pct:7.2f}")
            else:
                other_calls += calls
                other_tot += tot
        lines.append(f"{'(all non-cs kernels: torch input generation / copies)':90s} {other_calls:6d} {other_tot:14.0f}")
        if "--calls" in sys.argv:   # every dispatch of the kernels whose name contains the given text, in launch order (us)
            pat = sys.argv[sys.argv.index("--calls") + 1]
            try:
                rows = c.execute("select name, start, end from kernels order by start").fetchall()
                lines.append(f"# dispatches of *{pat}* in launch order (us)")
                for name, st, en in rows:
                    if pat in name:
                        lines.append(f"{short(name):70s} {(en - st) / 1000.0:12.1f}")
            except sqlite3.Error as e:
                lines.append(f"# (--calls: no per-dispatch view in this database: {e})")
    else:
        cur = c.execute("select * from counters_collection limit 1")
        cols = [d[0] for d in cur.description]
        lines.append(f"# rocprofv3 --pmc summary of {db}: mean counter value per dispatch")
        kcol = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x and "kernel" in x][0]
        ccol = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x and "name" in x][0]
        vcol = "value" if "value" in cols else "counter_value"
        q = f"select {kcol},{ccol},count(*),avg({vcol}),sum({vcol}) from counters_collection group by {kcol},{ccol}"
        for k, cn, n, avg, tot in c.execute(q).fetchall():
            if "cs::" in k:
                lines.append(f"{short(k):70s} {cn:28s} n={n:5d} mean={avg:18.1f}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

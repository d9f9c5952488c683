#!/usr/bin/env python3
"""Dump the per-kernel summary (calls, total/avg duration, %) of a rocprofv3 rocpd .db, and of PMC
counter rows if present, as plain text - what gets committed under profiles/."""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    print(f"# rocprofv3 kernel summary of {path}")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>7}  kernel")
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{calls:7d} {total:12.1f} {avg:10.3f} {pct:7.2f}  {name}")
    if "counters_collection" in tabs:
        cur = c.execute("select * from counters_collection limit 1")
        cols = [d[0] for d in cur.description]
        if "counter_name" in cols and "value" in cols and "kernel_name" in cols:
            print("\n# PMC counters: mean per dispatch")
            q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                 "group by kernel_name, counter_name order by kernel_name, counter_name")
            for k, n, v, cnt in c.execute(q):
                print(f"{v:18.1f}  n={cnt:5d}  {n:32s} {k[:80]}")


if __name__ == "__main__":
    main(sys.argv[1])

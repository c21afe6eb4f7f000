#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) as text for profiles/.
usage: rocprof_summary.py <results.db> [title]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    con = sqlite3.connect(db)
    cur = con.cursor()
    print("# %s" % title)
    print("# source: rocprofv3 --kernel-trace --stats (rocpd database), durations in ns")
    print("%-60s %8s %16s %14s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        # top_kernels reports microseconds
        print("%-60s %8d %16.0f %14.0f %8.2f" % (name[:60], calls, total * 1e3, avg * 1e3, pct))
    print()
    print("# per-dispatch rows of the search kernel")
    print("%-10s %12s %10s %8s %8s %8s %8s" % ("dispatch", "duration_ns", "grid_x", "wg_x", "lds", "vgpr", "sgpr"))
    for row in cur.execute("select dispatch_id,duration,grid_x,workgroup_x,lds_size,vgpr_count,sgpr_count from kernels "
                           "where name like '%sg_%' order by dispatch_id"):
        print("%-10d %12d %10d %8d %8d %8d %8d" % row)


if __name__ == "__main__":
    main()

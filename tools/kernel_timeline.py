#!/usr/bin/env python3
"""Timeline of the last call's kernels from a rocprofv3 --kernel-trace database: start / end (us, relative), queue, name.
usage: kernel_timeline.py <results.db> [n_last]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute("select start, end, %s, name from kernels where name like '%%sg%%' order by start" % (qcol or "0")))
rows = rows[-n:]
t0 = rows[0][0]
for s, e, q, name in rows:
    print("%10.1f %10.1f %8.1f  q%-4s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name[:60]))

"""Per-kernel mean counter values of the rocprofv3 result databases under a directory (one sub-directory per pass)."""
import glob
import os
import sqlite3
import sys

for sub in sorted(glob.glob(os.path.join(sys.argv[1], "*"))):
    f = glob.glob(os.path.join(sub, "**", "*results.db"), recursive=True)
    if not f:
        continue
    d = sqlite3.connect(f[0])
    try:
        rows = d.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events "
                         "where name like '%sgcn::%' group by name, counter_name order by avg(duration) desc limit 12").fetchall()
    except sqlite3.OperationalError:
        rows = []
    if not rows:
        try:
            for r in d.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 8"):
                print("%-22s %-90s calls %5d avg_us %10.1f pct %5.1f" % (os.path.basename(sub), r[0][:90], r[1], r[3], r[4]))
        except sqlite3.OperationalError:
            pass
    for r in rows:
        print("%-22s %-70s %-28s n %4d mean %16.1f ns %10.0f" % (os.path.basename(sub), r[0][:70], r[1], r[2], r[3], r[4]))

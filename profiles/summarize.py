#!/usr/bin/env python
"""Turn rocprofv3 (ROCm 7.2, rocpd sqlite output) result databases into the small text
summaries committed under profiles/.

    python profiles/summarize.py gpurun_out/prof r01

expects  <dir>/trace/*_results.db      rocprofv3 --kernel-trace --stats -- python bench.py ...
         <dir>/pmc_fetch/*_results.db  rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ...
         <dir>/pmc_write/*_results.db  rocprofv3 --pmc WRITE_SIZE --kernel-trace -- ...
         <dir>/pmc_l2/*_results.db     rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -- ...
(each PMC pass is its own run, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).
"""
import glob
import os
import sqlite3
import sys


def db(path):
    f = glob.glob(os.path.join(path, "**", "*results.db"), recursive=True)   # <dir>/<host>/<pid>_results.db
    return sqlite3.connect(f[0]) if f else None


def main(src, tag):
    out_dir = os.path.dirname(os.path.abspath(__file__))
    lines = []
    t = db(os.path.join(src, "trace"))
    if t:
        lines.append("== rocprofv3 --kernel-trace --stats : per-kernel totals (microseconds: the rocpd top_kernels view)")
        lines.append("%-100s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for r in t.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 15"):
            lines.append("%-100s %8d %14d %12.0f %7.2f" % (r[0][:100], r[1], r[2], r[3], r[4]))
    for sub, note in (("pmc_fetch", "FETCH_SIZE is in KiB; on gfx950 it reports 1/2 of a wide coalesced read "
                                    "stream (x2 correction, calibrate on the copy kernel in the same trace)"),
                      ("pmc_write", "WRITE_SIZE is in KiB"),
                      ("pmc_l2", "L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)")):
        d = db(os.path.join(src, sub))
        if not d:
            continue
        lines.append("")
        lines.append("== rocprofv3 --pmc (%s) : per-kernel mean counter value per dispatch   [%s]" % (sub, note))
        lines.append("%-100s %-14s %6s %16s %12s" % ("kernel", "counter", "n", "mean_value", "mean_ns"))
        q = ("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events "
             "group by name, counter_name order by avg(duration) desc limit 12")
        for r in d.execute(q):
            lines.append("%-100s %-14s %6d %16.1f %12.0f" % (r[0][:100], r[1], r[2], r[3], r[4]))
    # machine-readable traffic record of the dominant (longest) sgcn kernel, read by bench.py
    try:
        import json
        def top(dbh, counter):
            q = ("select name, avg(counter_value), avg(duration) from pmc_events where counter_name=? "
                 "and name like '%sgcn::%' group by name order by avg(duration) desc limit 1")
            return dbh.execute(q, (counter,)).fetchone()
        f = top(db(os.path.join(src, "pmc_fetch")), "FETCH_SIZE")
        w = top(db(os.path.join(src, "pmc_write")), "WRITE_SIZE")
        l2 = db(os.path.join(src, "pmc_l2"))
        hit = l2.execute("select avg(counter_value) from pmc_events where counter_name='TCC_HIT_sum' and name=?", (f[0],)).fetchone()[0]
        mis = l2.execute("select avg(counter_value) from pmc_events where counter_name='TCC_MISS_sum' and name=?", (f[0],)).fetchone()[0]
        bj = json.load(open(os.path.join(src, "bench_trace.json")))
        rec = {"kernel": f[0], "fetch_kib_raw": f[1], "write_kib": w[1],
               "fetch_bytes_corrected": f[1] * 1024 * 2, "write_bytes": w[1] * 1024,
               "hbm_bytes_per_launch": f[1] * 1024 * 2 + w[1] * 1024,
               "kernel_launches_per_spmm": bj["config"].get("kernel_launches_per_spmm", 1),
               "hbm_bytes_per_spmm": (f[1] * 1024 * 2 + w[1] * 1024) * bj["config"].get("kernel_launches_per_spmm", 1),
               "l2_hit_rate": hit / (hit + mis), "ns_per_launch_profiled": f[2],
               "nnz": bj["config"]["nnz"], "d": bj["config"]["d"],
               "note": "FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section; "
                       "calibrated on the 566.6 MB copy kernel of the same trace); WRITE_SIZE as reported"}
        json.dump(rec, open(os.path.join(out_dir, "%s_traffic.json" % tag), "w"), indent=1)
        lines.append("")
        lines.append("== dominant kernel traffic per launch: fetch %.2f GB (corrected) + write %.2f GB, L2 hit %.3f; "
                     "%d launches per SpMM -> %.2f GB per SpMM"
                     % (rec["fetch_bytes_corrected"] / 1e9, rec["write_bytes"] / 1e9, rec["l2_hit_rate"],
                        rec["kernel_launches_per_spmm"], rec["hbm_bytes_per_spmm"] / 1e9))
    except Exception as e:   # partial profile directories are fine
        lines.append("(no traffic record: %s)" % e)
    path = os.path.join(out_dir, "%s_rocprof_summary.txt" % tag)
    open(path, "w").write("\n".join(lines) + "\n")
    print(path)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

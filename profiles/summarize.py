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
    notes = {"pmc_fetch": "FETCH_SIZE is in KiB; on gfx950 it reports 1/2 of a wide coalesced read "
                          "stream (x2 correction, calibrate on the copy kernel in the same trace)",
             "pmc_write": "WRITE_SIZE is in KiB",
             "pmc_l2": "L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)"}
    subs = ["pmc_fetch", "pmc_write", "pmc_l2"] + sorted(
        os.path.basename(x) for x in glob.glob(os.path.join(src, "pmc_*"))
        if os.path.isdir(x) and os.path.basename(x) not in notes)
    counters = {}
    for sub in subs:
        note = notes.get(sub, "separate run")
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
            if "sgcn::cs_spmm" in r[0] or "sgcn::spmm_seg" in r[0]:
                counters.setdefault(r[0], {})[r[1]] = {"mean_per_dispatch": r[3], "mean_ns": r[4], "dispatches": r[2]}
    # machine-readable traffic record of the dominant (longest) sgcn kernel, read by bench.py
    try:
        import json
        def top(dbh, counter):
            # the kernel the bench TIMES: most time in total (dispatches x mean duration), not the longest single dispatch --
            # since round 6 the same command also runs the row-gather kernel four times for the setup report
            q = ("select name, avg(counter_value), avg(duration) from pmc_events where counter_name=? "
                 "and name like '%sgcn::%' group by name order by sum(duration) desc limit 1")
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
    if counters:
        import json as _json
        _json.dump(counters, open(os.path.join(out_dir, "%s_counters.json" % tag), "w"), indent=1)
        for k, c in counters.items():
            g = lambda n: c.get(n, {}).get("mean_per_dispatch")       # noqa: E731
            lines.append("")
            lines.append("== derived, %s" % k[:90])
            ns = next(iter(c.values()))["mean_ns"]
            lines.append("   mean dispatch %.1f us" % (ns / 1e3))
            cyc = g("GRBM_GUI_ACTIVE")             # shader-clock cycles of the dispatch (per XCD instance)
            if cyc:
                lines.append("   %.0f shader cycles (%.2f GHz)" % (cyc, cyc / ns))
            if g("TCC_BUSY_avr") is not None and cyc:
                lines.append("   L2 (TCC) busy %.1f %% of the dispatch  [TCC_BUSY_avr = %.0f cycles, mean over the L2 channels]"
                             % (100.0 * g("TCC_BUSY_avr") / cyc, g("TCC_BUSY_avr")))
            if g("TA_BUSY_avr") is not None and cyc:
                lines.append("   texture-address units (TA) busy %.1f %%  [TA_BUSY_avr = %.0f cycles]; of that stalled by the cache on "
                             "addresses %.1f %%, on data %.1f %% of the dispatch"
                             % (100.0 * g("TA_BUSY_avr") / cyc, g("TA_BUSY_avr"),
                                100.0 * (g("TA_ADDR_STALLED_BY_TC_CYCLES_sum") or 0) / 256 / cyc,
                                100.0 * (g("TA_DATA_STALLED_BY_TC_CYCLES_sum") or 0) / 256 / cyc))
            if g("TCP_PENDING_STALL_CYCLES_sum") is not None and cyc:
                lines.append("   vector L1 (TCP) stalled on pending misses %.1f %% of the dispatch (mean over 256 CUs)"
                             % (100.0 * g("TCP_PENDING_STALL_CYCLES_sum") / 256 / cyc))
            if g("TCC_EA0_RDREQ_LEVEL_sum") and g("TCC_EA0_RDREQ_sum"):
                lines.append("   mean fabric read latency seen by the L2: %.0f cycles"
                             % (g("TCC_EA0_RDREQ_LEVEL_sum") / g("TCC_EA0_RDREQ_sum")))
            if g("TCP_UTCL1_TRANSLATION_MISS_sum") is not None:
                lines.append("   L1 TLB misses per dispatch: %.0f (negligible)" % g("TCP_UTCL1_TRANSLATION_MISS_sum"))
            if g("TCP_TCC_READ_REQ_sum") and g("TCP_TOTAL_CACHE_ACCESSES_sum"):
                lines.append("   vector L1: %.3g tag lookups, %.3g read requests to L2 (x 128 B = %.2f GB per dispatch)"
                             % (g("TCP_TOTAL_CACHE_ACCESSES_sum"), g("TCP_TCC_READ_REQ_sum"),
                                g("TCP_TCC_READ_REQ_sum") * 128 / 1e9))
            if g("TCP_TCC_READ_REQ_LATENCY_sum") and g("TCP_TCC_READ_REQ_sum"):
                lines.append("   mean L1->L2 read latency %.0f cycles" % (g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")))
            if g("SQ_WAIT_INST_ANY") and g("SQ_WAVE_CYCLES"):
                lines.append("   waves waiting on an instruction dependency %.1f %% of wave-cycles; issuing VMEM %.1f %%, VALU %.1f %%, scalar %.1f %%"
                             % (100.0 * g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
                                100.0 * (g("SQ_ACTIVE_INST_VMEM") or 0) / g("SQ_WAVE_CYCLES"),
                                100.0 * (g("SQ_ACTIVE_INST_VALU") or 0) / g("SQ_WAVE_CYCLES"),
                                100.0 * (g("SQ_ACTIVE_INST_SCA") or 0) / g("SQ_WAVE_CYCLES")))
            if g("TCC_EA0_RDREQ_sum") and g("TCC_REQ_sum"):
                lines.append("   L2: %.3g requests, %.3g fabric read requests, tag stalls %.3g, DRAM-credit stalls %.3g"
                             % (g("TCC_REQ_sum"), g("TCC_EA0_RDREQ_sum"), g("TCC_TAG_STALL_sum") or 0,
                                g("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum") or 0))
    path = os.path.join(out_dir, "%s_rocprof_summary.txt" % tag)
    open(path, "w").write("\n".join(lines) + "\n")
    print(path)
    print("\n".join(lines))


def traffic_from_counters(counters_path, bench_path, out_path):
    """The traffic record of the bench's timed kernel from a ``*_counters.json`` + the bench line of the traced run (the
    databases of the r62 passes stayed on the GPU box; their per-kernel means are in the counters file)."""
    import json
    c = json.load(open(counters_path))
    bj = json.loads(open(bench_path).read().strip().splitlines()[-1])
    name = max(c, key=lambda k: c[k]["FETCH_SIZE"]["dispatches"] * c[k]["FETCH_SIZE"]["mean_ns"])
    k = c[name]
    f, w = k["FETCH_SIZE"]["mean_per_dispatch"], k["WRITE_SIZE"]["mean_per_dispatch"]
    hit, mis = k["TCC_HIT_sum"]["mean_per_dispatch"], k["TCC_MISS_sum"]["mean_per_dispatch"]
    nl = bj["config"].get("kernel_launches_per_spmm", 1)
    rec = {"kernel": name, "fetch_kib_raw": f, "write_kib": w, "fetch_bytes_corrected": f * 1024 * 2, "write_bytes": w * 1024,
           "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024, "kernel_launches_per_spmm": nl,
           "hbm_bytes_per_spmm": (f * 1024 * 2 + w * 1024) * nl, "l2_hit_rate": hit / (hit + mis),
           "ns_per_launch_profiled": k["FETCH_SIZE"]["mean_ns"], "nnz": bj["config"]["nnz"], "d": bj["config"]["d"],
           "note": "FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; "
                   "rebuilt from the per-kernel means of the separate --pmc passes (profiles/summarize.py --from-counters)"}
    if "per_gpu" in bj["config"] and "rank 0 rows" in str(bj["config"]["per_gpu"]):
        import re
        m = re.search(r"(\d+) nnz", bj["config"]["per_gpu"])
        if m:
            rec["nnz"] = int(m.group(1))            # a block of a sharded graph is identified by ITS nonzeros (bench.py)
    json.dump(rec, open(out_path, "w"), indent=1)
    return rec


if __name__ == "__main__":
    if sys.argv[1] == "--from-counters":
        print(traffic_from_counters(sys.argv[2], sys.argv[3], sys.argv[4]))
    else:
        main(sys.argv[1], sys.argv[2])

#!/usr/bin/env python
"""Build and run profiles/gather_ceiling.hip on the GPU; write the result table as JSON.

    python profiles/gather_ceiling.py [out.json]        (default: gpurun_out/gather_ceiling.json)

The table's best "window"/"hit"/"miss" rates for the shipped access shape (1,216 B per nonzero)
are what bench.py reports `roofline.frac_of_gather_ceiling` against (profiles/gather_ceiling.json).
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gather_ceiling.json")
    exe = os.path.join(HERE, "gather_ceiling.bin")
    src = os.path.join(HERE, "gather_ceiling.hip")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", src, "-o", exe])
    res = subprocess.run([exe], stdout=subprocess.PIPE, text=True, check=True)
    rows = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    best = {}
    for r in rows:
        if r["bytes_per_row"] == 1216:
            k = r["regime"]
            if k not in best or r["TBps"] > best[k]["TBps"]:
                best[k] = r
    doc = {"what": "pure row gather into VGPRs, no FMA (profiles/gather_ceiling.hip), S-Reddit operand "
                   "(232,965 rows x 2,432 B), 4,096 rows per wave",
           "best_1216B": best, "rows": rows}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for r in rows:
        if r["regime"].startswith("mixed"):
            print(r["regime"] + ": hit-only %.3f ms (%.1f TB/s), miss-only %.3f ms (%.1f TB/s), both in one launch %.3f ms "
                  "(sum %.3f, max %.3f)" % (r["ms_hit_only"], r["TBps_hit_only"], r["ms_miss_only"], r["TBps_miss_only"],
                                            r["ms"], r["ms_hit_only"] + r["ms_miss_only"], max(r["ms_hit_only"], r["ms_miss_only"])))
            continue
        print("%-7s %-13s %5d B  U=%-2d %-7s wps=%d  %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU" % (
            r["regime"], r["shape"], r["bytes_per_row"], r["U"], r["policy"], r["waves_per_simd"], r["ms"], r["TBps"],
            r["B_per_clk_per_CU"]))


if __name__ == "__main__":
    main()

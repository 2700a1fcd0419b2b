#!/bin/bash
# Profile one bench.py configuration of the column-sweep SpMM under rocprofv3 (ROCm 7.2):
#   profiles/profile_cs.sh <tag> [bench.py args...]
# -> gpurun_out/prof_<tag>/{trace,pmc_*}/...   then   python profiles/summarize.py gpurun_out/prof_<tag> <tag>
# One run for --kernel-trace --stats, and ONE RUN PER COUNTER GROUP (the guide's rule: PMC passes are
# their own runs; never combined with sys/hip/hsa tracing).  The pace is fixed (--tune cs_pace=...) so that
# no autotune launches pollute the per-kernel averages.
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
BENCH="python $PWD/bench.py --no-epoch --no-cpu-baseline --steps 10 --warmup 2 $*"
cd /tmp && export TMPDIR=/tmp
# every rocprofv3 run is bounded: a counter set the hardware cannot collect makes the tool abort and then
# hang in its own finalisation (seen with four TA counters in one pass)
want() { [ -z "${ONLY:-}" ] || [[ " $ONLY " == *" $1 "* ]]; }
if want trace; then
    echo "== trace"; timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || tail -3 "$OUT/trace.err"
fi
run_pmc() { name=$1; shift; want $name || return 0; echo "== $name: $*"; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -- $BENCH > /dev/null 2> "$OUT/$name.err" || { echo "   FAILED"; grep -m2 -E "error code|exceeds" "$OUT/$name.err"; }; }
run_pmc pmc_fetch FETCH_SIZE
run_pmc pmc_write WRITE_SIZE
run_pmc pmc_l2 TCC_HIT_sum TCC_MISS_sum
run_pmc pmc_tcc TCC_REQ_sum TCC_READ_sum TCC_BUSY_avr TCC_TAG_STALL_sum
run_pmc pmc_tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
run_pmc pmc_tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run_pmc pmc_tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_UTCL1_TRANSLATION_MISS_sum
run_pmc pmc_ta TA_BUSY_avr TA_TA_BUSY_sum
run_pmc pmc_ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run_pmc pmc_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM
run_pmc pmc_sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
cd "$OLDPWD"
python profiles/summarize.py "gpurun_out/prof_$TAG" "$TAG" > "$OUT/summary.log" 2>&1
tail -60 "$OUT/summary.log"

#!/bin/bash
# PMC passes (one run per counter group, never with other tracing) over profiles/lds_run.py:
#   profiles/lds_pmc.sh <tag> [lds_run.py args]   ->  gpurun_out/pmc_<tag>/table.txt
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
RUN="python $PWD/profiles/lds_run.py $*"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- $RUN > "$OUT/run.log" 2> "$OUT/trace.err"
pass() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -- $RUN > /dev/null 2> "$OUT/$name.err" || echo "$name FAILED"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS
cd "$ROOT"
python profiles/pmc_table.py "$OUT" > "$OUT/table.txt" 2>&1
cat "$OUT/run.log" | tail -2
cat "$OUT/table.txt"

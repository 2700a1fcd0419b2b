#!/bin/bash
# Counter passes (one run per group) of the step's first dense layer alone: profiles/gemm_probe.py --only dense0
#   profiles/gemm_pmc.sh  ->  gpurun_out/gemm_pmc.txt
set -u
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/gemm_pmc
rm -rf "$out"; mkdir -p "$out"
cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -- python "$repo/profiles/gemm_probe.py" --only dense0 > /dev/null 2> "$out/$name.err" || echo "$name FAILED"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
run sq3 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD
run sq4 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cd "$repo"
python - "$out" > gpurun_out/gemm_pmc.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "gemm_kernel" not in k and "splitk" not in k:
            continue
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-34s n=%4d  mean %.4g" % (c, len(v), sum(v) / len(v)))
PY
cat gpurun_out/gemm_pmc.txt

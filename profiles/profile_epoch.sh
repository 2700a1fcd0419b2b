#!/bin/bash
# rocprofv3 kernel trace of the CVD+PP training epoch (profiles/epoch_profile.py) + the per-queue chain summary.
#   profiles/profile_epoch.sh <tag>     ->  gpurun_out/<tag>_train_epoch_kernels.txt
set -u
tag=${1:-ep}
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$out" -- python "$repo/profiles/epoch_profile.py" 3 > "$out/run.log" 2>&1
cd "$repo"
f=gpurun_out/${tag}_train_epoch_kernels.txt
python profiles/epoch_profile.py --summarize "$out" 894 > $f 2>&1
python profiles/epoch_profile.py --gaps "$out" >> $f 2>&1
python profiles/epoch_profile.py --timeline "$out" >> $f 2>&1
python profiles/epoch_profile.py --stepgaps "$out" >> $f 2>&1
tail -3 "$out/run.log" >> $f
rm -rf "$out"

R=$PWD
OUT=$R/gpurun_out/prof4
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-epoch --no-backward --tune cs_pace=280"
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.err
PYTHONPATH=$R timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -- $CMD > /dev/null 2> $OUT/f.err
PYTHONPATH=$R timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -- $CMD > /dev/null 2> $OUT/w.err
PYTHONPATH=$R timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_l2 -- $CMD > /dev/null 2> $OUT/l.err
cd $R
python profiles/summarize.py $OUT r04 | tail -8
cp profiles/r04_rocprof_summary.txt profiles/r04_traffic.json gpurun_out/
cp $OUT/bench_trace.json gpurun_out/r04_bench_under_rocprof.json
rm -rf $OUT

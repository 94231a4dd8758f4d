cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in coinrun bigfish; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_$g -o t -- python $R/bench.py --game $g --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/tr_$g.log 2>&1
tail -1 $R/gpurun_out/tr_$g.log | cut -c1-200
find $R/gpurun_out/tr_$g -name "*kernel_stats.csv" | head -1 | xargs cat | head -8
done
rm -rf $R/gpurun_out/tr_coinrun $R/gpurun_out/tr_bigfish

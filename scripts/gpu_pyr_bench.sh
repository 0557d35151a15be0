# pyramid ResNet bench (bench.py --config pyramid) under rocprofv3 --kernel-trace --stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/pyr
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -o pyr -- python3 bench.py --config pyramid > $O/bench_pyramid.json 2> $O/bench_pyramid.err
cat $O/bench_pyramid.json
f=$(find $O/prof_b -name "*kernel_stats.csv" | head -1)
cp "$f" $O/pyramid_bench_kernel_stats.csv
head -22 $O/pyramid_bench_kernel_stats.csv | cut -c1-200
rm -rf $O/prof_b

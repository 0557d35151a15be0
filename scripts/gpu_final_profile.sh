# the bench line and the rocprofv3 kernel summary of the SAME command (default bench.py), for profiles/
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python3 $R/bench.py > $R/gpurun_out/bench_final.json 2> $R/gpurun_out/bench_final.err
tail -1 $R/gpurun_out/bench_final.json | cut -c1-200
cp $R/gpurun_out/prof_final/*/*kernel_stats.csv $R/gpurun_out/final_kernel_stats.csv
head -6 $R/gpurun_out/final_kernel_stats.csv | cut -c1-150

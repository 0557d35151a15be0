cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pyr
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pyr/prof -o pyr -- python3 scripts/gpu_pyramid.py pyramid 2000 16384 0.85 > gpurun_out/pyr/prof_run.log 2>&1
tail -12 gpurun_out/pyr/prof_run.log | grep -v warn
f=$(ls gpurun_out/pyr/prof/*/pyr_kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(find gpurun_out/pyr/prof -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -c1-260
cp "$f" gpurun_out/pyr/pyr_kernel_stats.csv
rm -rf gpurun_out/pyr/prof

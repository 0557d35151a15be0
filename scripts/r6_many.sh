#!/bin/bash
# round 6, item 5: timelines of run_core on one million 500-bp records (and the 10 000-contig file) with the 500-bp model;
# kernel + memory-copy statistics of the million-record run; cProfile of its calling thread
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6many
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/r5_e2e_timeline.py many 4 > $O/timeline_many.log 2>&1
python3 $R/scripts/r5_e2e_timeline.py 10k 3 > $O/timeline_10k.log 2>&1
python3 $R/scripts/r4_e2e_many.py > $O/cprofile_many.log 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/prof -- python3 $R/scripts/r5_e2e_timeline.py many 2 > $O/prof_many.log 2>&1
cp $O/prof/*/*kernel_stats.csv $O/many_kernel_stats.csv 2>/dev/null
cp $O/prof/*/*memory_copy_stats.csv $O/many_memcpy_stats.csv 2>/dev/null
cp $O/prof/*/*domain_stats.csv $O/many_domain_stats.csv 2>/dev/null
rm -rf $O/prof
grep "==" $O/timeline_many.log $O/timeline_10k.log
grep -A1 "==" $O/timeline_many.log | tail -4

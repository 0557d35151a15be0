# round 5: timelines of short-forward end-to-end runs (10 000-contig FASTA and one million 500-bp records), repeat scan
# beside the forward (default) and in front of it (A/B)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5b; exec > gpurun_out/r5b/run.log 2>&1
python -m pytest tests/test_gpu_cli.py -q 2>&1 | tail -3
python scripts/r5_e2e_timeline.py 10k 4 2>&1 | grep "==\|@"
JAEGER_SCAN_FIRST=1 python scripts/r5_e2e_timeline.py 10k 4 2>&1 | grep "==\|@"
python scripts/r5_e2e_timeline.py many 3 2>&1 | grep "==\|@"
JAEGER_SCAN_FIRST=1 python scripts/r5_e2e_timeline.py many 3 2>&1 | grep "==\|@"

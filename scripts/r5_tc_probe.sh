# round 5: Toom-Cook F(2,5) main-loop probe (scripts/ubench/tc_probe.hip), three runs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5t
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/tc_probe.hip -o /tmp/tc_probe 2>&1 | grep -v warning | head -5
for r in 1 2 3; do echo "== run $r"; timeout 120 /tmp/tc_probe; done > gpurun_out/r5t/tc_probe.txt 2>&1
cat gpurun_out/r5t/tc_probe.txt

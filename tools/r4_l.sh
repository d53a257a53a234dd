cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
AVAMD_GEMM_V7=0 timeout -k 10 300 python tools/v7_ab.py save /tmp/v4_out.pt > gpurun_out/r4l/save.log 2>&1; echo "save(v4) rc=$?"
timeout -k 10 300 python tools/v7_ab.py compare /tmp/v4_out.pt > gpurun_out/r4l/compare.log 2>&1; echo "compare rc=$?"; tail -1 gpurun_out/r4l/compare.log
for i in 1 2; do timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4l/probe_$i.log 2>&1; echo "probe rc=$?"; done
cat gpurun_out/r4l/probe_2.log

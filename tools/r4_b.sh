cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
AVAMD_GEMM_V7=0 timeout -k 10 300 python tools/v7_ab.py save /tmp/v4_out.pt > gpurun_out/r4b/save.log 2>&1; echo "save rc=$?"
timeout -k 10 300 python tools/v7_ab.py compare /tmp/v4_out.pt > gpurun_out/r4b/compare.log 2>&1; echo "compare rc=$?"
tail -2 gpurun_out/r4b/compare.log
for i in 1 2; do
AVAMD_GEMM_V7=0 timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4b/probe_v4_$i.log 2>&1; echo "v4 rc=$?"
AVAMD_GEMM_V7=1 timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4b/probe_v7_$i.log 2>&1; echo "v7 rc=$?"
done
paste gpurun_out/r4b/probe_v4_2.log gpurun_out/r4b/probe_v7_2.log | cut -c1-220

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
AVAMD_GEMM_V7=0 timeout -k 10 300 python tools/v7_ab.py save /tmp/v4_out.pt > gpurun_out/r4i/save.log 2>&1; echo "save(v4) rc=$?"
timeout -k 10 300 python tools/v7_ab.py compare /tmp/v4_out.pt > gpurun_out/r4i/compare.log 2>&1; echo "compare(v7 noskip vs v4) rc=$?"; tail -2 gpurun_out/r4i/compare.log
for i in 1 2; do
AVAMD_LIB=tools/_bin/libavhip_rolled.so timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4i/probe_branchy_$i.log 2>&1; echo "branchy rc=$?"
timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4i/probe_noskip_$i.log 2>&1; echo "noskip rc=$?"
done
paste gpurun_out/r4i/probe_branchy_2.log gpurun_out/r4i/probe_noskip_2.log | cut -c1-220
timeout -k 10 250 python tools/pace_probe.py 2>&1 | grep -v amdgpu | cut -c1-150

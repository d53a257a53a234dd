set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "fast or linear or gelu_gradient or layout" > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r4a/pytest.log
AVAMD_GEMM_V7=0 timeout -k 10 300 python tools/v7_ab.py save /tmp/v4_out.pt > gpurun_out/r4a/save.log 2>&1; echo "save rc=$?"
timeout -k 10 300 python tools/v7_ab.py compare /tmp/v4_out.pt > gpurun_out/r4a/compare.log 2>&1; echo "compare rc=$?"
tail -15 gpurun_out/r4a/compare.log
for i in 1 2; do
AVAMD_GEMM_V7=0 timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4a/probe_v4_$i.log 2>&1; echo "v4 rc=$?"
AVAMD_GEMM_V7=1 timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4a/probe_v7_$i.log 2>&1; echo "v7 rc=$?"
done
paste gpurun_out/r4a/probe_v4_1.log gpurun_out/r4a/probe_v7_1.log | cut -c1-220

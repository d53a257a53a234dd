cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
AVAMD_GEMM_V7=0 timeout -k 10 300 python tools/v7_ab.py save /tmp/v4_out.pt > gpurun_out/r4e/save.log 2>&1; echo "save rc=$?"
timeout -k 10 300 python tools/v7_ab.py compare /tmp/v4_out.pt > gpurun_out/r4e/compare.log 2>&1; echo "compare rc=$?"
grep -v "^checked" gpurun_out/r4e/compare.log | tail -30
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "fast or linear or gelu_gradient or layout" > gpurun_out/r4e/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4e/pytest.log
for i in 1 2; do
AVAMD_GEMM_V7_SPLIT=0 timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4e/probe_nosplit_$i.log 2>&1; echo "nosplit rc=$?"
timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4e/probe_split_$i.log 2>&1; echo "split rc=$?"
done
paste gpurun_out/r4e/probe_nosplit_2.log gpurun_out/r4e/probe_split_2.log | cut -c1-220

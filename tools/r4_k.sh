cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
AVAMD_GEMM_V7=0 timeout -k 10 300 python tools/v7_ab.py save /tmp/v4_out.pt > gpurun_out/r4k/save.log 2>&1; echo "save(v4) rc=$?"
timeout -k 10 300 python tools/v7_ab.py compare /tmp/v4_out.pt > gpurun_out/r4k/compare.log 2>&1; echo "compare(default vs v4) rc=$?"; tail -1 gpurun_out/r4k/compare.log
AVAMD_GEMM_V4_BM=224 timeout -k 10 300 python tools/v7_ab.py compare /tmp/v4_out.pt > gpurun_out/r4k/compare224.log 2>&1; echo "compare(bm 224 forced vs v4) rc=$?"; tail -1 gpurun_out/r4k/compare224.log
AVAMD_GEMM_V4_BM=208 timeout -k 10 300 python tools/v7_ab.py compare /tmp/v4_out.pt > gpurun_out/r4k/compare208.log 2>&1; echo "compare(bm 208 forced vs v4) rc=$?"; tail -1 gpurun_out/r4k/compare208.log
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "fast or linear or gelu_gradient or layout" > gpurun_out/r4k/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4k/pytest.log
for i in 1 2; do
AVAMD_GEMM_V4_BM=256 timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4k/probe_256_$i.log 2>&1; echo "256 rc=$?"
timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4k/probe_auto_$i.log 2>&1; echo "auto rc=$?"
done
paste gpurun_out/r4k/probe_256_2.log gpurun_out/r4k/probe_auto_2.log | cut -c1-220

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_stochastic_gpu.py -x -q -k "attention" > gpurun_out/r4g/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4g/pytest.log
for i in 1 2; do
AVAMD_ATTN_BWD2=0 timeout -k 10 120 python tools/attn_bwd_ab.py 2>&1 | grep -v amdgpu | sed 's/^/two-phase: /'
timeout -k 10 120 python tools/attn_bwd_ab.py 2>&1 | grep -v amdgpu | sed 's/^/one-pass : /'
done
ATTN_T=49 AVAMD_ATTN_BWD2=0 timeout -k 10 120 python tools/attn_bwd_ab.py 2>&1 | grep -v amdgpu | sed 's/^/two-phase: /'
ATTN_T=49 timeout -k 10 120 python tools/attn_bwd_ab.py 2>&1 | grep -v amdgpu | sed 's/^/one-pass : /'

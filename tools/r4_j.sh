cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -x -q -k "fast or linear or gelu_gradient or layout or conv or visual" > gpurun_out/r4j/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4j/pytest.log
AVAMD_LIB=tools/_bin/libavhip_branchy.so timeout -k 10 300 python tools/v7_ab.py save /tmp/br_out.pt > gpurun_out/r4j/save.log 2>&1; echo "save(branchy) rc=$?"
timeout -k 10 300 python tools/v7_ab.py compare /tmp/br_out.pt > gpurun_out/r4j/compare.log 2>&1; echo "compare(new vs branchy) rc=$?"; tail -2 gpurun_out/r4j/compare.log
for i in 1 2; do
AVAMD_LIB=tools/_bin/libavhip_branchy.so timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4j/probe_branchy_$i.log 2>&1; echo "branchy rc=$?"
timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4j/probe_new_$i.log 2>&1; echo "new rc=$?"
done
paste gpurun_out/r4j/probe_branchy_2.log gpurun_out/r4j/probe_new_2.log | cut -c1-220
for v in 1 0 1 0; do
  if [ $v = 1 ]; then lib=multimodal-av-model_amd/libavhip.so; else lib=tools/_bin/libavhip_branchy.so; fi
  AVAMD_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --single-variant --steps 20 --warmup 5 > gpurun_out/r4j/bench_$v.log 2>&1 || { echo bench failed; tail -3 gpurun_out/r4j/bench_$v.log; exit 1; }
  grep '^{' gpurun_out/r4j/bench_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('new' if $v else 'branchy', d['value'], 'utt/s', d['ms_per_step'], 'ms; GEMM family', r['achieved'], 'TF/s', r['avg_launch_us'], 'us', r['frac'])"
done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
timeout -k 10 300 python tools/v7_ab.py save /tmp/ref_out.pt > gpurun_out/r4h/save.log 2>&1; echo "save(unrolled) rc=$?"
AVAMD_LIB=tools/_bin/libavhip_rolled.so timeout -k 10 300 python tools/v7_ab.py compare /tmp/ref_out.pt > gpurun_out/r4h/compare.log 2>&1; echo "compare(rolled vs unrolled) rc=$?"; tail -2 gpurun_out/r4h/compare.log
for i in 1 2; do
AVAMD_LIB=tools/_bin/libavhip_rolled.so timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4h/probe_rolled_$i.log 2>&1; echo "rolled rc=$?"
timeout -k 10 200 python tools/epi_probe.py > gpurun_out/r4h/probe_unrolled_$i.log 2>&1; echo "unrolled rc=$?"
done
paste gpurun_out/r4h/probe_rolled_2.log gpurun_out/r4h/probe_unrolled_2.log | cut -c1-220

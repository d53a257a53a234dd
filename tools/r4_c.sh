cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --single-variant --steps 20 --warmup 5 > gpurun_out/r4c/$name.log 2>&1 || { echo "$name failed"; tail -5 gpurun_out/r4c/$name.log; return 1; }
  python - "$name" <<'PY'
import json, sys
l = [x for x in open(f"gpurun_out/r4c/{sys.argv[1]}.log") if x.startswith("{")][-1]
d = json.loads(l); r = d.get("roofline") or {}
print(sys.argv[1], d["value"], "utt/s", d["ms_per_step"], "ms; roofline", r.get("achieved"), "TF/s", r.get("avg_launch_us"), "us", flush=True)
PY
}
for i in 1 2; do
run v7_g256_$i AVAMD_GEMM_V7=1 && run v7_g128_$i AVAMD_GEMM_V7=1 AVAMD_GEMM_V7_G=128 && run v4_$i AVAMD_GEMM_V7=0 && run v7_g192_$i AVAMD_GEMM_V7=1 AVAMD_GEMM_V7_G=192 || exit 1
done

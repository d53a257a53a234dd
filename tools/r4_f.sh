cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
for cfg in "fp16 64 --loss-scaling" "bf16 64" "fp16 128 --loss-scaling" "bf16 128 --loss-scaling"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --precision $1 --batch $2 $3 --single-variant --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r4f/${1}_b$2.log 2>&1; echo "$cfg rc=$?"
  grep '^{' gpurun_out/r4f/${1}_b$2.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['dtype'], d['value'], d['ms_per_step'], d['config']['final_loss'], d['config'].get('loss_scaling'), d['roofline']['achieved'] if d.get('roofline') else None)"
done

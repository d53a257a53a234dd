cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
timeout -k 10 400 python bench.py --h2d --single-variant --no-cpu-baseline --no-probe --steps 20 --warmup 5 > gpurun_out/r4d/h2d.log 2>&1; echo "h2d rc=$?"
grep '^{' gpurun_out/r4d/h2d.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['h2d']))"
timeout -k 10 400 python bench.py --force-dp --single-variant --no-cpu-baseline --no-probe --steps 20 --warmup 5 > gpurun_out/r4d/forcedp.log 2>&1; echo "dp rc=$?"
grep '^{' gpurun_out/r4d/forcedp.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['config']['data_parallel']))"

# the other BASELINE configs on one box (parity-test sizes, not bench lines): as executed / deterministic; config 5 also under the GradScaler law
for args in "--batch 2 --seconds 1" "--batch 32" "--batch 8 --seconds 15" "--batch 128 --steps 10 --warmup 3" "--batch 128 --steps 10 --warmup 3 --loss-scaling" "--batch 128 --steps 10 --warmup 3 --loss-scaling --precision fp16"; do
  timeout -k 10 400 python bench.py $args --no-cpu-baseline --no-probe 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$args', '| valid', d['valid'], '| as_executed', d['value'], 'utt/s', d['ms_per_step'], 'ms', d['config']['step_tflops'], 'TF/s loss', d['config']['final_loss'], d['config'].get('loss_scaling'), '| deterministic', d['other_variant']['value'], d['other_variant']['ms_per_step'], 'ms')
"
done

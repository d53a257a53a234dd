"""Count the ATen operators (and their call stacks' top frames) executed per training step: finds stray copies / fills on the host side."""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_profile.py")).read().split("for _ in range(3):")[0])
for _ in range(3):
    t.train_step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
        t.train_step(batch)
torch.cuda.synchronize()
evs = [e for e in prof.events() if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::zero_", "aten::fill_", "aten::zeros", "aten::to", "aten::_to_copy")]
cnt = collections.Counter()
for e in evs:
    st = [f for f in (e.stack or []) if "multimodal-av-model_amd" in f or "torch/autograd" in f]
    cnt[(e.name, str(e.input_shapes)[:60], st[0][-90:] if st else "?")] += 1
for k, v in cnt.most_common(45):
    print(v / 2, k)

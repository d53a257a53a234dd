"""BiLSTM layer timing in isolation (T=100, B=$LSTM_B (64), H=512, bf16): per-step kernels vs the persistent kernel."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("multimodal-av-model_amd.ops"); L = importlib.import_module("multimodal-av-model_amd._lib")
T, B, H = 100, int(os.environ.get("LSTM_B", "64")), 512
dt = torch.bfloat16
gx = torch.randn(T, B, 2, 4 * H, device="cuda")
whh = (torch.randn(2, 4 * H, H, device="cuda") / 22).to(dt)
whhT = whh.transpose(1, 2).contiguous()
hseq = torch.empty(T, B, 2 * H, device="cuda", dtype=dt); cseq = torch.empty(T, B, 2, H, device="cuda")
gates = torch.empty(T, B, 2, 4 * H, device="cuda", dtype=dt); dg = torch.empty_like(gates); dc = torch.empty(2, B, H, device="cuda")
dout = torch.randn(B, T, 2 * H, device="cuda"); cnt = torch.empty(L.LSTM_COUNTER_INTS, dtype=torch.int32, device="cuda")
st = ops.stream()
def steps_f():
    for s in range(T):
        L.check(L.lib().av_lstm_fwd_step(ops.ptr(gx), ops.ptr(whh), ops.ptr(hseq), ops.ptr(cseq), ops.ptr(gates), None, 1, T, B, H, s, st))
def steps_b():
    for s in range(T):
        L.check(L.lib().av_lstm_bwd_step(ops.ptr(dout), 0, T * 2 * H, 2 * H, ops.ptr(dg), ops.ptr(whhT), ops.ptr(gates), ops.ptr(cseq), ops.ptr(dc), 1, T, B, H, s, st))
def pers_f():
    L.check(L.lib().av_lstm_fwd_layer(ops.ptr(gx), ops.ptr(whh), ops.ptr(hseq), ops.ptr(cseq), ops.ptr(gates), None, ops.ptr(cnt), T, B, H, st))
def pers_b():
    L.check(L.lib().av_lstm_bwd_layer(ops.ptr(dout), 0, T * 2 * H, 2 * H, ops.ptr(dg), ops.ptr(whhT), ops.ptr(gates), ops.ptr(cseq), ops.ptr(dc), ops.ptr(cnt), T, B, H, st))
for name, fn in (("fwd steps", steps_f), ("fwd persistent", pers_f), ("bwd steps", steps_b), ("bwd persistent", pers_b)):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:16s} {e0.elapsed_time(e1) / 5 * 1000 / T:7.2f} us/step   flag={int(cnt[2]) if 'pers' in name else '-'}", flush=True)

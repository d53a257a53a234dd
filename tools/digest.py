"""Digest of an end-of-round measurement batch (tools/r4_final.sh <tag>): bench line, per-step kernel-time categories, PMC summaries."""
import csv, json, os, sys
tag = sys.argv[1]
d = json.load(open(f"gpurun_out/{tag}/bench_line.json"))
r = d["roofline"]; at = d["attention"]; ov = d["other_variant"]; cb = d["cpu_baseline"]
print(f"value {d['value']} utt/s, {d['ms_per_step']} ms/step, final_loss {d.get('final_loss')}, valid {d.get('valid')}")
print("step keys:", {k: d[k] for k in d if k.startswith(("algorithmic", "step_", "encoder", "audio", "launch"))})
print(f"other_variant {ov['value']} utt/s {ov['ms_per_step']} ms loss {ov['final_loss']}")
print("roofline:", {k: r[k] for k in r if k not in ("kernel", "traffic_note")})
for k, v in at.items():
    print(" attention", k, {a: v[a] for a in ("avg_launch_us", "launches_per_step", "tflops", "frac_of_mfma_peak", "gb_per_s") if a in v})
print("cpu_baseline:", cb)
f = f"gpurun_out/{tag}/prof/kernel_stats.csv"
if os.path.exists(f):
    rows = list(csv.DictReader(open(f)))
    steps = 17                                               # 10 timed + 3 warm-up + 4 steps of the roofline probe leg
    tot = sum(int(x["TotalDurationNs"]) for x in rows) / steps / 1e6
    nl = sum(int(x["Calls"]) for x in rows) / steps
    print(f"kernel time per profiled step {tot:.1f} ms, {nl:.0f} launches per step (17 profiled steps: 10 timed + 3 warm-up + 4 probe)")
    def cat(n):
        if "gemm_nt_bf16_v7" in n or "gemm_nt_bf16_v4_kernel<false, false>" in n: return "NT GEMM (v7/v4)"
        if "gemm_nt_bf16_v4_kernel<false, true>" in n or "gemm_nt_bf16_kernel<128, false, true" in n or "sum_slices" in n: return "dW (k-major) + slice sums"
        if "v4_kernel<true" in n or "gemm_nt_bf16_kernel<128, true" in n or "gemm_nt_bf16_kernel<64, true" in n: return "lip conv GEMMs"
        if "gemm" in n: return "other GEMM"
        if "attn" in n or "xattn" in n: return "attention"
        if "ln_" in n: return "LayerNorm"
        if "conv3d" in n or "conv3x3" in n or "bn_" in n or "pool" in n or "avgpool" in n: return "lip conv3d / layer1 / BN passes"
        if "lstm" in n: return "BiLSTM"
        if "at::" in n or "rocclr" in n: return "torch / runtime helpers"
        return "other"
    c = {}
    for x in rows:
        c[cat(x["Name"])] = c.get(cat(x["Name"]), 0) + int(x["TotalDurationNs"]) / steps / 1e6
    print({k: round(v, 2) for k, v in sorted(c.items(), key=lambda kv: -kv[1])})
    for x in rows[:14]:
        print(f"  {int(x['Calls'])/steps:7.1f}/step {float(x['AverageNs'])/1e3:8.1f} us  {x['Name'][:100]}")

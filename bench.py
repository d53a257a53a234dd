"""Benchmark of the hot path: utterances/sec of the full audio-visual CTC training step (fwd + bwd + Adam).

  python bench.py --gpus N --steps K --warmup W            (N > 1: one rank per GPU over RCCL)

Workload = the configuration BASELINE.json's metric is quoted on: batch 64 per GPU x 4 s clips (64 000 samples @16 kHz + 100 lip
frames 96x96 per speaker), wav2vec2-large architecture, random-init weights, synthetic data, bf16 MFMA compute with fp32 master
weights; N > 1 keeps the per-GPU batch (weak scaling) and all-reduces the 63.8 M trainable gradients over RCCL
(``python bench.py --gpus N`` starts its own N ranks; under torch.distributed.run it is one of them).
Headline ``value`` = the step AS THE REFERENCE EXECUTES IT: two wav2vec2 passes per step (model/trainer.py:94-95) with the HF-default
train-mode regularisers (dropout 0.1, LayerDrop 0.1, SpecAugment 0.05).  ``other_variant`` = the deterministic parity configuration
(all regularisers 0), where the two passes are bit-identical and the encoder runs once.
Inputs are resident in HBM before the timed region.  One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import importlib
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "multimodal-av-model_amd"


def imp(sub):
    return importlib.import_module(PKG + "." + sub)


def flops_per_utt(cfg, T_audio, T_v, audio_passes, layers=None):
    """SURVEY §8(d) algorithmic FLOPs (2*MAC) per utterance, fwd + needed bwd, for the variant executed.  ``layers`` = encoder layers
    actually executed per step, summed over the audio passes: (forward, backward without weight gradients, backward with weight
    gradients) - counted by the model from its LayerDrop draws (hf:774-789); None = no layer dropped."""
    L = T_audio
    conv = 0.0
    cin = 1
    for k, s, c in zip(cfg["conv_kernel"], cfg["conv_stride"], cfg["conv_dim"]):
        L = (L - k) // s + 1
        conv += 2.0 * k * cin * c * L
        cin = c
    T = L
    H, I, nl = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    lin = (8.0 * H * H + 4.0 * H * I) * T
    att = 4.0 * T * T * H
    kp, G = cfg["num_conv_pos_embeddings"], cfg["num_conv_pos_embedding_groups"]
    if layers is None:
        layers = (audio_passes * nl, audio_passes * 14, audio_passes * 4)
    n_f, n_b, n_bt = layers
    A0 = conv + 2.0 * cin * H * T + 2.0 * kp * (H // G) * H * T          # feature extractor + projection + positional conv
    enc = n_f * (lin + att) + n_b * (lin + 2 * att) + n_bt * (2 * lin + 2 * att)
    V = 695.2e6 * T_v
    F = T_v * (2 * 512 ** 2 + 2 * 1024 * 512 + 8 * 512 ** 2 + 2 * 512 ** 2 + 2 * 4 * 512 * (1024 + 1536) * 2 + 2 * 1024 * 800) + 4.0 * T_v ** 2 * 512
    # with two audio passes the frozen, dropout-free conv feature extractor runs ONCE (model/trainer.py: shared between the passes):
    # count what is executed
    return audio_passes * A0 - (audio_passes - 1) * conv + enc + 2 * V + 2 * F + 2 * 2 * F


def cpu_baseline(cfg, seconds):
    """The CPU oracle (checked against the reference in the build container) timed on this box's host cores."""
    from oracle import av_oracle as O
    init = imp("utils.init"); synth = imp("dataset.synthetic")
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))         # a 1-GPU box owns a 16-core CPU share
    B = 2
    batch = synth.make_batch(B, seconds, seed=42)
    sds = [init.visual_state_dict(), init.w2v2_state_dict(cfg), init.fusion_state_dict(512, cfg["hidden_size"], 512),
           init.decoder_state_dict(1024, 800)]
    proj = init.projection_params(cfg["hidden_size"])
    st = {}
    O.train_step(*sds, cfg, batch, proj, st, dedup_audio=False)         # warm-up
    t0 = time.time()
    n = 2
    for _ in range(n):
        O.train_step(*sds, cfg, batch, proj, st, dedup_audio=False)     # as the reference executes it: two audio passes
    dt = (time.time() - t0) / n
    return {"value": round(B / dt, 4), "unit": "utterances/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"B={B} x {seconds:g} s clips, {n} fp32 steps after 1 warm-up, two audio passes as the reference runs it"}


HF_REGULARIZERS = dict(hidden_dropout=0.1, attention_dropout=0.1, activation_dropout=0.1, feat_proj_dropout=0.0, layerdrop=0.1,
                       mask_time_prob=0.05, mask_time_length=10, mask_time_min_masks=2)
NO_REGULARIZERS = dict(hidden_dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0,
                       mask_time_prob=0.0)


def build_trainer(batch_size, seconds, precision, dev, rank=0, reducer=None, lambda_=0.1, no_pair=False, no_side_stream=False,
                  loss_scaling=False):
    """The benchmark's model (wav2vec2-large + ResNet-18 lip encoder + fusion + CTC head, seeded random weights, the reference's freeze
    policy main.py:26-31,100-106), its trainer and one HBM-resident synthetic batch with the host-side metadata merged in."""
    init = imp("utils.init"); synth = imp("dataset.synthetic"); enc = imp("model.encoder"); fm = imp("model.fusion_module")
    dm = imp("model.decoder"); tr = imp("model.trainer"); tok = imp("utils.tokenizer")
    imp("precision").set_precision(precision)
    cfg = dict(init.W2V2_LARGE)
    ve = enc.VisualEncoder(); ve.load_state_dict(init.visual_state_dict())
    for p in ve.parameters():
        p.requires_grad = False
    ae = enc.AudioEncoder(dict(cfg), freeze=True)
    for n, p in ae.model.named_parameters():
        p.requires_grad = any(f"encoder.layers.{i}." in n for i in range(6, 10))
    fu = fm.CrossAttentionFusion(512, cfg["hidden_size"], 512); fu.load_state_dict(init.fusion_state_dict(512, cfg["hidden_size"], 512))
    de = dm.CTCDecoder(1024, 800, 3); de.load_state_dict(init.decoder_state_dict(1024, 800))
    t = tr.MultimodalTrainer(ve, ae, fu, de, tok.SyntheticTokenizer(800), learning_rate=1e-4, device=dev, lambda_=lambda_,
                             audio_passes=None, reducer=reducer, pair_batched=not no_pair, visual_side_stream=not no_side_stream,
                             loss_scaling=loss_scaling)
    if no_side_stream:
        imp("model.w2v2").PASS_STREAMS = False
    t.fixed_projection = init.projection_params(cfg["hidden_size"])      # identical on every rank (SURVEY §8e caveat 4)
    t.visual_encoder.train(); t.audio_encoder.train(); t.fusion_module.train(); t.decoder1.train()
    cpu_batch = synth.make_batch(batch_size, seconds, seed=42 + rank)
    T_enc = int(imp("model.w2v2").conv_out_lengths(cfg, cpu_batch["audio"].shape[1]))
    batch = {k: v.to(dev) for k, v in cpu_batch.items()}
    batch.update(t.host_metadata(cpu_batch, T_enc))        # class counts + CTC lengths from the host copy: no device read-back in the step
    batch.update(_T_audio=cpu_batch["audio"].shape[1], _T_v=cpu_batch["lip1"].shape[1], _T_enc=T_enc)
    return t, batch, cfg


def spawn_ranks(n: int) -> int:
    """``python bench.py --gpus N`` as typed: this process has not touched the GPU yet, so it starts N fresh ranks under
    torch.distributed.run (one per GPU, RCCL), relays their output and returns their exit code."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def attention_report(records, steps):
    """SURVEY §8(d) attention-kernel report: per kernel (K7 = wav2vec2 self-attention, head_dim 64; K17 = fusion cross-attention,
    head_dim 128), forward and backward: average launch time (events on the launch stream), algorithmic TFLOP/s against the dense
    bf16 MFMA peak and algorithmic GB/s against the HBM peak."""
    out = {}
    names = {"fwd64": "K7_w2v2_self_attention_fwd", "bwd64": "K7_w2v2_self_attention_bwd",
             "fwd128": "K17_fusion_cross_attention_fwd", "bwd128": "K17_fusion_cross_attention_bwd",
             "blockfwd": "K17_fused_block_fwd", "blockbwd": "K17_attention_core_bwd"}
    for tag in sorted({r[2] for r in records}):
        rs = [r for r in records if r[2] == tag]
        ms = sum(r[0].elapsed_time(r[1]) for r in rs)
        fl = sum(r[3] for r in rs); by = sum(r[4] for r in rs)
        tf = fl / (ms * 1e-3) / 1e12
        gb = by / (ms * 1e-3) / 1e9
        out[names.get(tag, tag)] = {"avg_launch_us": round(1e3 * ms / len(rs), 2), "launches_per_step": len(rs) // max(1, steps),
                                    "algorithmic_gflop_per_launch": round(fl / len(rs) / 1e9, 3), "tflops": round(tf, 1),
                                    "frac_of_mfma_peak": round(tf / 2500.0, 4), "algorithmic_mb_per_launch": round(by / len(rs) / 1e6, 2),
                                    "gb_per_s": round(gb, 1), "frac_of_hbm_peak": round(gb / 8000.0, 4)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE.json metric: batch 64)")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"], help="fp16 = the reference's GPU arithmetic (half MFMA operands, libavhip_f16.so); use with --loss-scaling")
    ap.add_argument("--variant", default="as_executed", choices=["as_executed", "deterministic"],
                    help="headline variant.  as_executed: two audio passes with HF-default dropout / LayerDrop / SpecAugment, what the reference's "
                         "train_epoch runs (model/trainer.py:94-95); deterministic: every regulariser 0 (the parity configuration), where the "
                         "two passes are bit-identical and the encoder runs once")
    ap.add_argument("--single-variant", action="store_true", help="time only the headline variant")
    ap.add_argument("--lambda", dest="lambda_", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --same-device rehearses N>1 on a 1-GPU box")
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--force-dp", action="store_true", help="build the process group and the gradient reducer even with one rank (RCCL path rehearsal)")
    ap.add_argument("--no-side-stream", action="store_true", help="one stream: visual encoder and second audio pass on the main stream (no overlap; profiling)")
    ap.add_argument("--no-pair", action="store_true", help="one fusion/decoder call per speaker, as the reference does")
    ap.add_argument("--loss-scaling", action="store_true", help="GradScaler law of the reference's GPU mode (model/trainer.py:40,121-123): BASELINE configs[4]")
    ap.add_argument("--h2d", action="store_true", help="also time the headline variant with the host->device leg of the reference's step (model/trainer.py:66-75) "
                                                       "INSIDE the timed region: batches in pinned host memory, copied one step ahead on a prefetch stream "
                                                       "(double buffer).  `value` stays the HBM-resident number; the line gains an `h2d` object")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))        # nothing above has initialised the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.same_device:
        local = 0
        # rehearsal only: the ranks share ONE GPU, where the persistent BiLSTM kernels of two processes cannot all be resident at once
        os.environ.setdefault("AVAMD_LSTM_PERSISTENT", "0")
    if (world > 1 or args.force_dp) and args.backend == "nccl":
        # The step's five streams want FOUR hardware queues to themselves (profiles/r04_hw_queues.txt: 5+ active queues cost 15 %, the ROCm default of 4 is
        # the optimum without a communicator); an initialised RCCL communicator takes two of the process's queues, and the one-rank leg measured
        # 66.3 / 65.6 / 64.7 / 74 ms per step at 4 / 5 / 6 / 8 queues against 64.4 without a communicator.  Read by the runtime when the device is first
        # touched (below); an explicit setting of the caller wins.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1 or args.force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo")

    ops = imp("ops"); L = imp("_lib")
    dp = imp("parallel.dp")
    reducer = dp.GradBucketReducer(always_collective=args.force_dp) if (world > 1 or args.force_dp) else None
    t, batch, cfg = build_trainer(args.batch, args.seconds, args.precision, dev, rank=rank, reducer=reducer, lambda_=args.lambda_,
                                  no_pair=args.no_pair, no_side_stream=args.no_side_stream, loss_scaling=args.loss_scaling)
    metric = f"utterances/sec (4 s clip, 25 fps 96x96 lip) at batch {args.batch}, full training step"
    dp_probe = None
    if world > 1 or args.force_dp:
        # evidence that the collective really runs over the ranks the line claims: observed world size and a one-time all-reduce checksum
        import torch.distributed as dist
        chk = torch.full((1024,), float(rank + 1), device=dev)
        dist.all_reduce(chk)
        torch.cuda.synchronize()
        dp_probe = {"rccl_ranks": dist.get_world_size(), "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "allreduce_checksum": float(chk.sum()),
                    "allreduce_checksum_expected": 1024.0 * world * (world + 1) / 2}
    ae = t.audio_encoder
    T_audio, T_v, T_enc = batch["_T_audio"], batch["_T_v"], batch["_T_enc"]

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    def set_variant(v):
        """The trainer picks the number of audio passes itself (two as soon as a regulariser is active in train mode)."""
        ae.model.cfg.update(HF_REGULARIZERS if v == "as_executed" else NO_REGULARIZERS)
        return 2 if v == "as_executed" else 1

    # Every leg (headline, other variant, roofline probe) starts from the SAME state: initial weights, BatchNorm running statistics, fresh
    # optimizer / scaler state.  No leg runs on the weights another leg left behind.
    state_tensors = [p for m in (t.audio_encoder, t.fusion_module, t.decoder1) for p in m.parameters() if p.requires_grad]
    state_tensors += [b for b in t.visual_encoder.buffers()]
    snap = [x.detach().clone() for x in state_tensors]

    def restore():
        with torch.no_grad():
            for x, s0 in zip(state_tensors, snap):
                x.copy_(s0)                                       # bumps the version counters: compute-dtype caches rebuild
        t.optimizer.reset_state()
        t.scaler = imp("optim").AvGradScaler(device=dev, enabled=args.loss_scaling)
        t.projection_layer = None

    def finite_state(loss_value):
        """A leg is a measurement only if it ends with a finite loss AND finite weights (one reduction per trainable tensor, once per leg)."""
        if not math.isfinite(loss_value):
            return False
        sums = torch.stack([p.detach().float().sum() for p in state_tensors])
        return bool(torch.isfinite(sums).all())

    pinned = None

    def timed(v, h2d=False):
        nonlocal pinned
        restore()
        passes = set_variant(v)
        torch.manual_seed(1234 + rank)
        import numpy as np
        np.random.seed(1234 + rank)
        m = ae.model
        pf = slot = None
        if h2d:
            # the reference's step starts with .to(device) of the batch (model/trainer.py:66-75): here the next batch is copied from pinned
            # host memory on a prefetch stream while the current step computes (dataset/prefetch.py), INSIDE the timed region
            pre = imp("dataset.prefetch")
            if pinned is None:
                pinned = pre.pin_batch({k: (x.cpu() if torch.is_tensor(x) and x.is_cuda else x) for k, x in batch.items()})
            pf = pre.DevicePrefetcher(dev)
            slot = pf.stage(pinned)

        def one_step():
            nonlocal slot
            if pf is None:
                return t.train_step(batch)
            cur = pf.get(slot)
            nxt = pf.stage(pinned)                                # batch i + 1 travels while step i computes
            o = t.train_step(cur)
            pf.release(slot)
            slot = nxt
            return o
        for _ in range(args.warmup):
            out = one_step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        m.layers_executed = m.layers_executed_bwd = m.layers_executed_bwd_tr = 0
        if reducer is not None:
            reducer.exposed_ms()                                  # drop the warm-up's event pairs
            reducer.timing = True
            reducer.flat_reduces = reducer.cat_reduces = 0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = one_step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt)
        lay = (m.layers_executed / args.steps, (m.layers_executed_bwd - m.layers_executed_bwd_tr) / args.steps, m.layers_executed_bwd_tr / args.steps)
        loss_v = float(out["total"].detach())
        dpi = None
        if reducer is not None:
            reducer.timing = False
            ex = reducer.exposed_ms()
            dpi = {"ranks": world, "backend": "rccl (torch.distributed nccl)" if args.backend == "nccl" else args.backend,
                   "gradient_bytes_per_step": int(sum(reducer.last_bucket_bytes)), "buckets_last_step_mb": [round(b / 1e6, 1) for b in reducer.last_bucket_bytes],
                   "buckets_reduced_in_place": reducer.flat_reduces, "buckets_packed_by_copy": reducer.cat_reduces,
                   "exposed_allreduce_ms_per_step": round(sum(ex) / max(1, len(ex)), 3),
                   "note": "buckets = decoder + fusion (issued when the wav2vec2 backward starts) and one per trainable wav2vec2 layer, all-reduced on a side "
                           "stream under the backward; exposed = time the main stream waited at the join before Adam (events on the main stream)"}
            if dp_probe is not None:
                dpi.update(dp_probe)
        ok = finite_state(loss_v)
        if world > 1:
            # every rank must take the SAME exit: a rank that left alone would strand its peers in the next collective
            import torch.distributed as dist
            fl_ = torch.tensor([1.0 if ok else 0.0], device=dev)
            dist.all_reduce(fl_, op=dist.ReduceOp.MIN)
            ok = bool(fl_.item() > 0.5)
        return dict(dt=dt, passes=passes, loss=loss_v, layers=lay, valid=ok, variant=v, dp=dpi,
                    scale=(t.scaler.get_scale(), t.scaler.steps_taken()) if args.loss_scaling else None,
                    h2d_bytes=(pf.bytes_last if pf is not None else None))

    def diverged(leg):
        if rank == 0:
            print(json.dumps({"metric": metric, "value": None,
                              "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "valid": False,
                              "status": "diverged", "config": {"variant": leg["variant"], "final_loss": None if not math.isfinite(leg["loss"]) else leg["loss"],
                                                               "note": "non-finite loss or weights at the end of the timed region: not a measurement"}}),
                  flush=True)
        if world > 1 or args.force_dp:
            import torch.distributed as dist
            dist.destroy_process_group()
        raise SystemExit(3)

    other = "deterministic" if args.variant == "as_executed" else "as_executed"
    head_leg = timed(args.variant)
    if not head_leg["valid"]:
        diverged(head_leg)
    second = None if args.single_variant else timed(other)
    if second is not None and not second["valid"]:
        diverged(second)
    h2d_leg = timed(args.variant, h2d=True) if args.h2d else None
    if h2d_leg is not None and not h2d_leg["valid"]:
        diverged(h2d_leg)
    # roofline leg: the SAME workload (same initial state, same seeds) for a few steps with per-launch events around the dominant kernel
    # and the attention launches.  The side streams (visual encoder, second audio pass) are switched off here so that the events bracket
    # only the kernel (with several streams the elapsed time between events includes the other streams' kernels); this is what
    # rocprofv3 reports as the duration of a kernel that runs alone.
    probe = attn = None
    if not args.no_probe and rank == 0 and world == 1:
        restore()
        set_variant(args.variant)
        torch.manual_seed(1234 + rank)
        import numpy as np
        np.random.seed(1234 + rank)
        t.visual_side_stream = False
        imp("model.w2v2").PASS_STREAMS = False                 # probe leg: one stream, so that the events bracket exactly one kernel
        t.train_step(batch)
        torch.cuda.synchronize()
        ops.GemmProbe.start(L.AV_BF16 if args.precision != "fp32" else L.AV_F32, L.A_ROWMAJOR, L.B_NK, True)
        ops.AttnProbe.start()
        probe_steps = min(3, args.steps)
        for _ in range(probe_steps):
            pout = t.train_step(batch)
        torch.cuda.synchronize()
        probe = ops.GemmProbe.stop()
        probe["steps"] = probe_steps
        probe["valid"] = finite_state(float(pout["total"].detach()))
        if os.environ.get("AVAMD_PROBE_SHAPES"):                # per-shape table of the probed family (tools / DESIGN only)
            agg = {}
            for r in probe["records"]:
                a = agg.setdefault(r[4], [0, 0.0, 0.0])
                a[0] += 1; a[1] += r[0].elapsed_time(r[1]); a[2] += r[2]
            with open(os.environ["AVAMD_PROBE_SHAPES"], "w") as f:
                f.write("M N K batch act bias R aux C2 drop conv stats | launches/step  us/launch  TF/s  ms/step\n")
                for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    f.write(" ".join(str(int(x)) for x in k) + f" | {a[0] / probe_steps:6.1f} {1000 * a[1] / a[0]:8.1f} {a[2] / a[1] / 1e9:7.1f} {a[1] / probe_steps:7.2f}\n")
        attn = attention_report(ops.AttnProbe.stop(), probe_steps)
        if not probe["valid"]:
            diverged(dict(variant=args.variant + " (roofline probe leg)", loss=float(pout["total"].detach())))

    if rank == 0:
        def line(leg):
            fl = flops_per_utt(cfg, T_audio, T_v, leg["passes"], leg["layers"])      # executed layers (LayerDrop draws are on the host), not expected ones
            utt = args.batch * world / (leg["dt"] / args.steps)
            out = {"value": round(utt, 3), "ms_per_step": round(1000.0 * leg["dt"] / args.steps, 3), "audio_passes": leg["passes"],
                   "wav2vec2_regularizers": "hf-defaults (dropout 0.1, LayerDrop 0.1, SpecAugment 0.05)" if leg["variant"] == "as_executed" else "off (deterministic parity configuration)",
                   "final_loss": round(leg["loss"], 4), "valid": leg["valid"],
                   "encoder_layers_executed_per_step": {"forward": round(leg["layers"][0], 2), "backward_dx_only": round(leg["layers"][1], 2),
                                                        "backward_dx_dw": round(leg["layers"][2], 2)},
                   "algorithmic_gflop_per_utt": round(fl / 1e9, 1), "step_tflops": round(fl * utt / 1e12, 1),
                   "step_frac_of_mfma_peak": round(fl * utt / 1e12 / peak, 4)}
            if leg.get("dp") is not None:
                out["data_parallel"] = leg["dp"]
            if leg["scale"] is not None:
                out["loss_scaling"] = {"law": "torch.amp.GradScaler (init 65536, x2 / 2000 clean steps, x0.5 on overflow, overflowing steps skipped)",
                                       "final_scale": leg["scale"][0], "optimizer_steps_taken": leg["scale"][1]}
            return out
        peak = 2500.0 if args.precision != "fp32" else 157.3
        head = line(head_leg)
        roof = None
        if probe and probe["records"]:
            tot_ms = sum(r[0].elapsed_time(r[1]) for r in probe["records"])
            tot_fl = sum(r[2] for r in probe["records"])
            tot_by = sum(r[3] for r in probe["records"])
            traffic = tsrc = None
            for cand in ("r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):       # separate rocprofv3 --pmc passes (see file)
                pmc = os.path.join(ROOT, "profiles", cand)
                if args.precision != "fp32" and os.path.exists(pmc):
                    tb = tl = 0                                                         # launch-weighted over the tilings of the family
                    for kname, kv in json.load(open(pmc))["kernels"].items():
                        if "gemm_nt_bf16_kernel<128, false, false, false>" in kname or "gemm_nt_bf16_v2_kernel" in kname or "gemm_nt_bf16_v4_kernel<false>" in kname or "gemm_nt_bf16_v7_kernel" in kname:
                            tb += kv["launches"] * (kv["fetch_bytes_per_launch"] + kv["write_bytes_per_launch"]); tl += kv["launches"]
                    traffic, tsrc = (round(tb / tl) if tl else None), cand
                    break
            n = len(probe["records"])
            ach = tot_fl / (tot_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": ("gemm_nt_bf16_v7_kernel (256x256x64 8-phase, persistent, register-direct epilogue; v4 = its one-tile-per-workgroup form for the classes v7 does not specialise) + the 128x128 / 256x128 tilings of the same family: row-major NT products = every nn.Linear forward, dX through cached W^T, strided conv1d" if args.precision != "fp32" else "gemm_kernel<float,128,0,0>"),
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                    "traffic_note": f"HBM-side bytes per launch from profiles/{tsrc} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)",
                    "algorithmic_bytes_per_launch": round(tot_by / n),
                    "launches_per_step": n // probe["steps"], "measured": "same workload from the same initial state and seeds, separate leg after the timed region, single stream", "avg_launch_us": round(1000.0 * tot_ms / n, 2),
                    "algorithmic_gflop_per_launch": round(tot_fl / n / 1e9, 3)}
        res = {"metric": metric, "value": head["value"],
               "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
               "config": {"workload": f"BASELINE metric configuration (= configs[3]'s per-GPU batch): batch {args.batch}/GPU x {args.seconds:g} s "
                                      f"(T_audio {T_audio}, T_enc {T_enc}, {T_v} lip frames 96x96 x 2 speakers), wav2vec2-large + ResNet-18 + fusion "
                                      "BiLSTM + CTC + contrastive, fwd+bwd+Adam, random-init weights",
                          "variant": args.variant, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                          "audio_passes": head["audio_passes"], "wav2vec2_regularizers": head["wav2vec2_regularizers"],
                          "lambda_contrastive": args.lambda_, "final_loss": head["final_loss"],
                          "encoder_layers_executed_per_step": head["encoder_layers_executed_per_step"],
                          "loss_scaling": head.get("loss_scaling"), "data_parallel": head.get("data_parallel"),
                          "algorithmic_gflop_per_utt": head["algorithmic_gflop_per_utt"],
                          "step_tflops": head["step_tflops"], "step_frac_of_mfma_peak": head["step_frac_of_mfma_peak"]},
               "valid": True, "roofline": roof}
        if second is not None:
            res["other_variant"] = dict(line(second), variant=other, steps=args.steps, warmup=args.warmup)
        if attn is not None:
            res["attention"] = attn
        if h2d_leg is not None:
            hl = line(h2d_leg)
            res["h2d"] = {"value": hl["value"], "ms_per_step": hl["ms_per_step"], "bytes_per_step": int(h2d_leg["h2d_bytes"]), "overlapped": True,
                          "exposed_ms": round(hl["ms_per_step"] - head["ms_per_step"], 3), "final_loss": hl["final_loss"],
                          "note": "same workload with the reference's host->device leg (model/trainer.py:66-75) inside the timed region: the batch lives in "
                                  "pinned host memory and batch i+1 is copied on a prefetch stream (double buffer, dataset/prefetch.py) while step i "
                                  "computes; exposed_ms = ms_per_step here minus the HBM-resident ms_per_step (`value` stays the resident number); "
                                  "tests/test_prefetch_gpu.py checks that prefetched steps equal resident steps bit for bit"}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, args.seconds)
        print(json.dumps(res), flush=True)
    if world > 1 or args.force_dp:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

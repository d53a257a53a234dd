"""cProfile of the host side of a few training steps at the bench configuration: where does the Python launch path spend its time?"""
import cProfile, importlib, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def imp(sub):
    return importlib.import_module("multimodal-av-model_amd." + sub)


init = imp("utils.init"); synth = imp("dataset.synthetic"); enc = imp("model.encoder"); fm = imp("model.fusion_module")
dm = imp("model.decoder"); tr = imp("model.trainer"); tok = imp("utils.tokenizer")
imp("precision").set_precision("bf16")
cfg = dict(init.W2V2_LARGE)
dev = "cuda:0"
ve = enc.VisualEncoder(); ve.load_state_dict(init.visual_state_dict())
for p in ve.parameters():
    p.requires_grad = False
ae = enc.AudioEncoder(dict(cfg), freeze=True)
for n, p in ae.model.named_parameters():
    p.requires_grad = any(f"encoder.layers.{i}." in n for i in range(6, 10))
fu = fm.CrossAttentionFusion(512, cfg["hidden_size"], 512); fu.load_state_dict(init.fusion_state_dict(512, cfg["hidden_size"], 512))
de = dm.CTCDecoder(1024, 800, 3); de.load_state_dict(init.decoder_state_dict(1024, 800))
t = tr.MultimodalTrainer(ve, ae, fu, de, tok.SyntheticTokenizer(800), learning_rate=1e-4, device=dev, lambda_=0.1, audio_passes=1)
t.fixed_projection = init.projection_params(cfg["hidden_size"])
t.visual_encoder.train(); t.audio_encoder.train(); t.fusion_module.train(); t.decoder1.train()
cpu_batch = synth.make_batch(32, 4.0, seed=42)
T_enc = int(imp("model.w2v2").conv_out_lengths(cfg, cpu_batch["audio"].shape[1]))
batch = {k: v.to(dev) for k, v in cpu_batch.items()}
batch.update(t.host_metadata(cpu_batch, T_enc))
for _ in range(3):
    t.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    t.train_step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue time per step {(t1 - t0) / 5 * 1e3:.2f} ms, wall per step {(t2 - t0) / 5 * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    t.train_step(batch)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue())

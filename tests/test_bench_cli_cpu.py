"""CPU: bench.py's command-line contract that needs no GPU - `--gpus N` typed directly starts N ranks under torch.distributed.run
(before anything touches the GPU) and relays their exit code; the FLOP accounting of the two variants."""
import importlib.util
import os
import sys

from conftest import ROOT, pkg


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpus_n_spawns_one_rank_per_gpu(monkeypatch):
    b = _load_bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert b.spawn_ranks(4) == 7                                  # the children's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


def test_flop_accounting_matches_survey_numbers():
    """4 s clips - 419 GF per utterance de-duplicated, 682 GF with two full audio passes (bench.py counts the shared conv
    feature extractor once: 662 GF)."""
    b = _load_bench()
    cfg = pkg("utils.init").W2V2_LARGE
    one = b.flops_per_utt(cfg, 64000, 100, 1) / 1e9
    two = b.flops_per_utt(cfg, 64000, 100, 2) / 1e9
    assert abs(one - 418.5) < 1.0 and abs(two - 662.2) < 1.0
    conv = sum(2.0 * k * ci * c * L for k, ci, c, L in zip(cfg["conv_kernel"], (1,) + tuple(cfg["conv_dim"][:-1]), cfg["conv_dim"],
                                                            (12799, 6399, 3199, 1599, 799, 399, 199))) / 1e9
    assert abs((two + conv) - 682.0) < 2.0


def test_flop_accounting_counts_executed_layers():
    """The LayerDrop draws are on the host, so bench.py prices the layers actually executed (forward / backward without / with weight
    gradients, summed over the passes), not an expectation: all layers executed = the plain two-pass figure; a dropped layer removes exactly
    its own forward and backward products."""
    b = _load_bench()
    cfg = pkg("utils.init").W2V2_LARGE
    full = b.flops_per_utt(cfg, 64000, 100, 2)
    assert b.flops_per_utt(cfg, 64000, 100, 2, (48, 28, 8)) == full
    H, I, T = cfg["hidden_size"], cfg["intermediate_size"], 199
    lin = (8.0 * H * H + 4.0 * H * I) * T
    att = 4.0 * T * T * H
    # one frozen layer (no backward) and one trainable layer (dX + dW) dropped in one pass each
    got = b.flops_per_utt(cfg, 64000, 100, 2, (46, 28, 7))
    assert abs((full - got) - ((lin + att) + (lin + att) + (2 * lin + 2 * att))) < 1.0
    # per-step averages need not be integers (LayerDrop over 30 timed steps)
    assert b.flops_per_utt(cfg, 64000, 100, 2, (43.0, 25.13, 7.13)) < full

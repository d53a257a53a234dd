"""CPU, world_size 2 over gloo: the bucketed gradient reducer and batch sharding used by the N>1 bench path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = importlib.import_module(PKG + ".parallel.dp"); synth = importlib.import_module(PKG + ".dataset.synthetic")
    red = dp.GradBucketReducer()
    assert red.world == world
    g = torch.Generator().manual_seed(100 + rank)
    grads = [torch.randn(7, 5, generator=g), torch.randn(11, generator=g), None, torch.randn(3, 2, 2, generator=g)]
    keep = [t.clone() for t in grads if t is not None]
    b1 = red.reduce_async(grads[:2]); b2 = red.reduce_async(grads[2:])          # two buckets in flight
    red.wait()
    views = b1 + b2
    # reference: explicit sum over ranks
    gather = [None] * world
    dist.all_gather_object(gather, [k.numpy() for k in keep])
    for i, v in enumerate(views):
        want = sum(torch.from_numpy(gather[r][i]) for r in range(world))
        assert torch.allclose(v, want, atol=1e-6), (rank, i)
        assert torch.allclose(v / world, want / world)
    # a bucket that already lives in one flat buffer (GradArena): reduced in place, views keep aliasing it, served once per step
    ar = dp.GradArena(align=4)
    assert ar.out("w", (3, 5), "cpu") is None and ar.out("b", (7,), "cpu") is None      # first backward: layout discovery
    ar.finalize(); ar.begin_step()
    w, b = ar.out("w", (3, 5), "cpu"), ar.out("b", (7,), "cpu")
    assert w is not None and b is not None and ar.owns(w) and ar.owns(b) and not ar.owns(torch.zeros(3))
    assert ar.out("w", (3, 5), "cpu") is None                                              # second request in the same step: not served
    w.copy_(torch.full((3, 5), float(rank + 1))); b.copy_(torch.arange(7.0) * (rank + 1))
    n_cat = red.cat_reduces
    red.reduce_flat(ar); red.wait()
    assert red.cat_reduces == n_cat and red.flat_reduces == 1
    assert torch.equal(w, torch.full((3, 5), 3.0)) and torch.equal(b, torch.arange(7.0) * 3)
    ar.begin_step()
    assert ar.out("w", (3, 5), "cpu").data_ptr() == w.data_ptr()                           # stable addresses from step to step
    # vector zone: zeroed by begin_step, accumulated into
    ar2 = dp.GradArena(align=4)
    assert ar2.out("bias", (6,), "cpu", vec=True) is None and ar2.out("w", (2, 3), "cpu") is None
    ar2.finalize(); ar2.begin_step()
    bv, wv = ar2.out("bias", (6,), "cpu", vec=True), ar2.out("w", (2, 3), "cpu")
    bv.add_(5.0); wv.fill_(7.0)
    ar2.begin_step()
    assert float(ar2.out("bias", (6,), "cpu", vec=True).abs().sum()) == 0.0 and float(ar2.out("w", (2, 3), "cpu").sum()) == 42.0
    batch = synth.make_batch(4, 0.1, seed=3)
    sh = dp.shard_batch(batch, rank, world)
    assert sh["audio"].shape[0] == 2 and torch.equal(sh["audio"], batch["audio"][rank * 2:(rank + 1) * 2])
    try:
        dp.shard_batch(synth.make_batch(3, 0.1, seed=3), rank, world)
        ok = False
    except ValueError:
        ok = True
    q.put((rank, ok))
    dist.destroy_process_group()


def test_bucket_allreduce_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, True), (1, True)]

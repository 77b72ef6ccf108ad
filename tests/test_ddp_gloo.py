"""N>1 path on CPU: world_size-2 gloo run of FlatBucketDDP over a stand-in flat model (same interface the native model
exposes: flat_params / flat_grads / grad_segments / _grad_sync), plus the meter all-reduce of the Runner (C5)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sota_imagenet_amd.parallel import FlatBucketDDP, plan_buckets


class FakeFlatModel(torch.nn.Module):
    """18 segments like the real executor; backward fills each segment with rank-dependent values and reports it."""

    def __init__(self, rank, kind="resnet50"):
        super().__init__()
        # the REAL executor's segment table (fc, 16 blocks, stem — a layout-only native context, no GPU), scaled down
        # 1:64 so the stand-in's flat arrays stay small; boundaries keep their order and relative sizes.  resnet50: the flat array is
        # laid out in backward order (segments ascend); bresnet50: in forward order (segments DESCEND through it)
        from sota_imagenet_amd.models import resnet50

        real = resnet50().grad_segments if kind == "resnet50" else resnet50(stem_type="deep", antialias=True, attn_type="eca", norm_layer="inplaceabn",
                                                                              norm_act="leaky_relu", weight_standardization=True).grad_segments
        assert len(real) == 18
        sizes = [max(16, (e - b) // 64) for b, e in real]
        descending = real[0][0] > real[-1][0]
        self._segments, off = [], 0
        for s in (reversed(sizes) if descending else sizes):
            self._segments.append((off, off + s))
            off += s
        if descending:
            self._segments.reverse()
        self.flat_params = torch.full((off,), float(rank + 1))
        self.flat_grads = torch.zeros(off)
        self._flat_buffers = torch.full((128,), float(10 * (rank + 1)))
        self._grad_sync = None
        self.rank = rank

    @property
    def grad_segments(self):
        return list(self._segments)

    def backward(self):
        for s, (b, e) in enumerate(self._segments):
            self.flat_grads[b:e] = (self.rank + 1) * (s + 1) + torch.arange(e - b) * 1e-3
            if self._grad_sync is not None:
                self._grad_sync(s, b, e)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, kind="resnet50"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = FakeFlatModel(rank, kind)
        ddp = FlatBucketDDP(m, bucket_cap_mb=32.0 / 64)  # the default 32 MiB cap at the same 1:64 scale
        # C2: everyone holds rank 0's parameters and buffers after construction
        ok = bool((m.flat_params == 1.0).all()) and bool((m._flat_buffers == 10.0).all())
        m.backward()
        exp = torch.zeros_like(m.flat_grads)
        for s, (b, e) in enumerate(m.grad_segments):
            exp[b:e] = sum((r + 1) * (s + 1) for r in range(world)) / world + torch.arange(e - b) * 1e-3
        ok = ok and torch.allclose(m.flat_grads, exp, rtol=0, atol=1e-5)
        nb = len(ddp.buckets)
        # no_sync() (accumulate_steps > 1): the backward inside issues no collective — gradients stay this rank's own
        with ddp.no_sync():
            m.backward()
        mine = torch.zeros_like(m.flat_grads)
        for s, (b, e) in enumerate(m.grad_segments):
            mine[b:e] = (rank + 1) * (s + 1) + torch.arange(e - b) * 1e-3
        ok = ok and torch.allclose(m.flat_grads, mine, rtol=0, atol=1e-6)
        m.backward()  # outside again: reduced
        ok = ok and torch.allclose(m.flat_grads, exp, rtol=0, atol=1e-5)
        # C5: Runner meter reduction
        from sota_imagenet_amd import fit_wrapper as fw

        r = fw.Runner(torch.nn.Linear(1, 1), None, None, callbacks=[])
        r.state.loss_meter.update(float(rank + 1), n=2)
        r._reduce_meters()
        ok = ok and abs(r.state.loss_meter.avg - 1.5) < 1e-12 and r.state.loss_meter.count == 4
        q.put((rank, ok, nb))
    finally:
        dist.destroy_process_group()


def test_flat_bucket_ddp_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(nb > 1 for _, _, nb in res)
    # the stand-in reduced the same number of buckets the native executor forms at the default cap
    from sota_imagenet_amd.models import resnet50

    assert all(nb == len(resnet50().bucket_plan(32.0)) for _, _, nb in res), res


def test_flat_bucket_ddp_world2_gloo_descending_segments():
    """the BResNet-50 executor's layout: forward-ordered flat array, backward segments descend through it"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "bresnet50")) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    from sota_imagenet_amd.bresnet import BResNet50

    m = BResNet50()
    assert all(nb == len(m.bucket_plan(32.0)) for _, _, nb in res), res
    segs = m.grad_segments
    assert segs[0][1] == m._nparam and segs[-1][0] == 0 and all(a[0] == b[1] for a, b in zip(segs, segs[1:]))
    for cap in (0.5, 2.0, 8.0, 32.0, 500.0):
        bk = m.bucket_plan(cap)
        assert bk == plan_buckets(segs, int(cap * (1 << 20) / 4)), cap
        assert bk[0][1] == m._nparam and bk[-1][0] == 0 and all(a[0] == b[1] for a, b in zip(bk, bk[1:]))


def test_bucket_plan_covers_every_segment_once():
    segs = [(0, 100), (100, 150), (150, 400), (400, 410), (410, 1000)]
    b = plan_buckets(segs, 200)
    assert b == [(0, 400, 2), (400, 1000, 4)]
    assert plan_buckets(segs, 10**9) == [(0, 1000, 4)]
    # the real layout: fc + 16 blocks + stem, 25 MB buckets
    from sota_imagenet_amd.models import resnet50

    m = resnet50()
    bk = plan_buckets(m.grad_segments, 25 * (1 << 20) // 4)
    assert bk[0][0] == 0 and bk[-1][1] == m.flat_grads.numel()
    for (b0, e0, _), (b1, e1, _) in zip(bk, bk[1:]):
        assert e0 == b1
    # the executor's own plan (mi355_resnet50_bucket_plan: what backward reduces once a communicator is attached) is the same
    for cap in (0.5, 8.0, 25.0, 32.0, 500.0):
        assert m.bucket_plan(cap) == plan_buckets(m.grad_segments, int(cap * (1 << 20) / 4)), cap

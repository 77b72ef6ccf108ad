"""Data parallelism for the flat-buffer model: bucketed gradient all-reduce over RCCL/xGMI on a side HIP stream.

On the GPU the collective lives INSIDE the C-ABI (csrc/comm.cpp: RCCL called directly): this class only bootstraps the
communicator (the ncclUniqueId travels through the process group the reference already creates at train.py:58-61),
broadcasts rank 0's state through it and attaches it to the model; every backward is then ONE native call that
reduces bucket after bucket on the communicator's own HIP stream.  torch.distributed is not on the data path.
The torch.distributed branch below serves the CPU stand-in of the world-size-2 gloo tests only.

Replaces `torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank])` at the reference's train.py:113-114
(process group from train.py:58-61, `init_process_group("nccl", "env://")` — on ROCm that backend IS RCCL).
What DDP does there, restated for one flat gradient array laid out in backward-completion order:
  * construction: rank-0 parameters and BN buffers are broadcast once (C2 of SURVEY §2.3);
  * backward: the native executor reports every finished segment (fc, then block by block, then the stem);
    consecutive segments form buckets of >= `bucket_cap_mb`; when a bucket's last segment has been ENQUEUED, an event
    is recorded on the compute stream, the side stream waits on it and issues ONE all-reduce (mean) over the bucket's
    contiguous slice — so the collective overlaps the rest of backward (C3);
  * after the last bucket the compute stream waits for the side stream, so optimizer.step() sees reduced gradients.
BN running statistics stay rank-local (the reference's per-forward buffer broadcast, C4, is dropped on purpose: it
does not influence training; rank 0's buffers are the ones checkpointed).
xGMI is point-to-point (7 links/GPU): few large buckets keep each ring step per-link efficient.
"""
import contextlib
import ctypes

import torch
import torch.distributed as dist
import torch.nn as nn


def plan_buckets(segments, cap_elems):
    """[(begin, end)] per backward segment -> [(begin, end, last_segment_index)] buckets of >= cap_elems elements; a bucket is the
    contiguous span of its segments (ascending for the ResNet-50 executor, whose flat array is laid out in backward order; descending for
    the BResNet-50 executor, laid out in forward order).  The last bucket (nothing left to overlap its all-reduce with) is cut once more:
    its trailing segments up to cap_elems // 8 form a bucket of their own, the part before them is reduced while those last blocks still
    compute (same rule as csrc/resnet_exec.cpp plan_buckets and csrc/bresnet_exec.cpp plan_bbuckets)."""
    def span(f, l):
        return (min(b for b, _ in segments[f:l + 1]), max(e for _, e in segments[f:l + 1]), l)

    buckets, firsts = [], []
    first, size = None, 0
    for i, (b, e) in enumerate(segments):
        if first is None:
            first, size = i, 0
        size += e - b
        if size >= cap_elems or i == len(segments) - 1:
            buckets.append(span(first, i))
            firsts.append(first)
            first = None
    tail_cap = cap_elems // 8
    if buckets and buckets[-1][1] - buckets[-1][0] > tail_cap:
        f, l = firsts[-1], buckets[-1][2]
        cut, tail = l + 1, 0
        for i in range(l, f, -1):
            b, e = segments[i]
            if tail + (e - b) > tail_cap:
                break
            tail += e - b
            cut = i
        if f < cut <= l:
            buckets[-1] = span(f, cut - 1)
            buckets.append(span(cut, l))
    return buckets


class FlatBucketDDP(nn.Module):
    def __init__(self, module, device_ids=None, bucket_cap_mb=32.0, process_group=None, broadcast=True):
        super().__init__()
        if not dist.is_initialized():
            raise RuntimeError("FlatBucketDDP needs torch.distributed.init_process_group first (train.py:61)")
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self._comm = None
        if not hasattr(module, "flat_grads"):  # a model of separate parameter tensors (bresnet.BResNet50)
            self._generic_init(bucket_cap_mb, broadcast)
            return
        flat = module.flat_grads
        self._cuda = flat.is_cuda
        self._buckets = plan_buckets(module.grad_segments, int(bucket_cap_mb * (1 << 20) / 4))
        self._by_last = {last: (b, e) for b, e, last in self._buckets}
        self._nseg = len(module.grad_segments)
        backend = dist.get_backend(process_group)
        self._avg = backend == "nccl"  # ReduceOp.AVG is an (R)CCL op; gloo sums and we scale
        if self._cuda and hasattr(module, "set_comm"):
            self._native_init(flat.device, bucket_cap_mb, broadcast)
            return
        if self._cuda:
            raise RuntimeError("FlatBucketDDP: a flat-array model on the GPU must carry its collective inside the executor (set_comm)")
        if broadcast:  # (CPU tensors from here on: the torch.distributed stand-in path of the gloo tests)
            self.broadcast_state()
        module._grad_sync = self._on_segment
        module._grad_sync_points = set(self._by_last)

    # ---- GPU: the collective inside the C-ABI ------------------------------------------------------------------
    def _native_init(self, device, bucket_cap_mb, broadcast):
        from . import native

        L = native.lib()
        if not L.mi355_comm_available():
            raise RuntimeError("FlatBucketDDP: librccl.so.1 not found (the MI355X data-parallel path has no fallback)")
        rank = dist.get_rank(self.group)
        uid = (ctypes.c_char * 128)()
        if rank == 0:
            native.check(L.mi355_comm_unique_id(uid))
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=0, group=self.group)  # bootstrap only: 128 bytes through the launcher's group
        comm = ctypes.c_void_p()
        native.check(L.mi355_comm_create(ctypes.byref(comm), box[0], self.world, rank, device.index or 0))
        self._comm = comm
        if broadcast:
            st = native.cur_stream()
            m = self.module
            native.check(L.mi355_comm_broadcast(comm, native.ptr(m.flat_params), m.flat_params.numel(), 0, st))
            native.check(L.mi355_comm_broadcast(comm, native.ptr(m._flat_buffers), m._flat_buffers.numel(), 0, st))
        self.module.set_comm(comm, bucket_cap_mb)
        native_plan = self.module.bucket_plan(bucket_cap_mb)
        assert native_plan == self._buckets, (native_plan, self._buckets)  # one plan, stated twice (C++ / plan_buckets)

    def _generic_init(self, bucket_cap_mb, broadcast):
        from . import native

        params = [p for p in self.module.parameters() if p.requires_grad]
        dev = params[0].device
        if not dev.type == "cuda":
            raise RuntimeError("FlatBucketDDP: the model must be on the GPU (no CPU data-parallel path)")
        L = native.lib()
        if not L.mi355_comm_available():
            raise RuntimeError("FlatBucketDDP: librccl.so.1 not found (the MI355X data-parallel path has no fallback)")
        rank = dist.get_rank(self.group)
        uid = (ctypes.c_char * 128)()
        if rank == 0:
            native.check(L.mi355_comm_unique_id(uid))
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=0, group=self.group)
        comm = ctypes.c_void_p()
        native.check(L.mi355_comm_create(ctypes.byref(comm), box[0], self.world, rank, dev.index or 0))
        self._comm = comm
        self._params = params
        self._gbuf = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
        self._buckets = [(0, self._gbuf.numel(), 0)]
        self._pending = False
        if broadcast:
            state = [t for t in self.module.state_dict().values() if t.is_floating_point()]
            flat = torch.cat([t.detach().reshape(-1).float() for t in state])
            native.check(L.mi355_comm_broadcast(comm, native.ptr(flat), flat.numel(), 0, native.cur_stream()))
            off = 0
            with torch.no_grad():
                for t in state:
                    t.copy_(flat[off: off + t.numel()].view(t.shape))
                    off += t.numel()
        for p in params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def _on_grad(self, _p):
        if getattr(self, "_skip_sync", False):
            return
        if not self._pending:
            self._pending = True
            torch.autograd.Variable._execution_engine.queue_callback(self._reduce_all)

    @torch.no_grad()
    def _reduce_all(self):
        from . import native

        self._pending = False
        off = 0
        for p in self._params:
            n = p.numel()
            if p.grad is not None:
                self._gbuf[off: off + n].view(p.grad.shape).copy_(p.grad)  # same memory order as the parameter (strides kept)
            off += n
        native.check(native.lib().mi355_comm_allreduce_mean(self._comm, native.ptr(self._gbuf), self._gbuf.numel(), native.cur_stream()))
        off = 0
        for p in self._params:
            n = p.numel()
            if p.grad is not None:
                p.grad.copy_(self._gbuf[off: off + n].view(p.grad.shape))
            off += n

    def __del__(self):
        try:
            if self._comm is not None:
                from . import native

                if hasattr(self.module, "set_comm"):
                    self.module.set_comm(None)
                native.lib().mi355_comm_destroy(self._comm)
                self._comm = None
        except Exception:
            pass

    # ---- C2: rank 0 -> everyone, once ----------------------------------------------------------------------
    def broadcast_state(self):
        with torch.no_grad():
            dist.broadcast(self.module.flat_params, 0, group=self.group)
            bufs = getattr(self.module, "_flat_buffers", None)
            if bufs is not None:
                dist.broadcast(bufs, 0, group=self.group)

    @property
    def buckets(self):
        return list(self._buckets)

    @contextlib.contextmanager
    def no_sync(self):
        """torch DDP's no_sync(): backwards inside keep their gradients rank-local (accumulate_steps > 1: every micro-step but
        the last); the first backward outside reduces the accumulated sums once."""
        if hasattr(self.module, "set_grad_sync"):
            self.module.set_grad_sync(False)
        else:
            self._skip_sync = True
        try:
            yield
        finally:
            if hasattr(self.module, "set_grad_sync"):
                self.module.set_grad_sync(True)
            else:
                self._skip_sync = False

    def comm_stats(self, reset=True):
        """[(kind, begin, end)] of every collective issued through the native communicator since the last reset
        (mi355_comm_stats: kind 0 bucket all-reduce over flat-gradient elements, 1 broadcast, 2 whole-buffer all-reduce)."""
        from . import native

        if self._comm is None:
            return []
        cap = 4096
        n, K, B, E = ctypes.c_int(), (ctypes.c_int * cap)(), (ctypes.c_size_t * cap)(), (ctypes.c_size_t * cap)()
        native.check(native.lib().mi355_comm_stats(self._comm, int(reset), cap, ctypes.byref(n), K, B, E))
        return [(K[i], B[i], E[i]) for i in range(min(n.value, cap))]

    # ---- C3: called by the model's backward right after a segment's kernels were enqueued ---------------------
    def _on_segment(self, seg, begin, end):
        if getattr(self, "_skip_sync", False):  # inside no_sync(): gradients stay rank-local
            return
        rng = self._by_last.get(seg)
        if rng is not None:
            # (CPU path only: on the GPU the executor reduces its own buckets, _native_init)
            self._reduce(self.module.flat_grads[rng[0]: rng[1]])

    def _reduce(self, view):
        if self._avg:
            dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            view.div_(self.world)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference's only parallelism is ``torch.nn.DataParallel`` (train.py:98-99): single process,
replicate/scatter/gather per step, gradients reduce-added onto GPU 0.  The MI355X design is one
process per GPU with identical replicas and local BatchNorm statistics (DataParallel's BN is per
replica too), and exactly one exchange per step: a sum all-reduce of the gradients divided by the
world size.  Gradients are packed into a few large flat buckets (xGMI is point-to-point, 7 links
x ~153 GB/s per GPU, so ring collectives are per-link bound: few large messages beat many small
ones), reduced asynchronously in reverse parameter order (the order backward produces them) and
copied back.  Works with any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in the CPU
tests).
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, model, world_size=None, bucket_bytes=64 << 20, group=None):
        self.group = group
        self.world = world_size if world_size is not None else dist.get_world_size(group)
        self.always = False    # True: issue the collectives even with one rank (single-GPU dry run of the RCCL path)
        params = [p for p in model.parameters() if p.requires_grad]
        params.reverse()  # backward finishes the last layers first
        self.buckets, cur, cur_bytes = [], [], 0
        for p in params:
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)

    def sync(self):
        """Average ``.grad`` of every parameter across ranks (missing grads count as zero).

        Per bucket: ONE pack kernel (torch.cat of the flattened grads), one asynchronous all-reduce,
        one scale, one multi-tensor copy back -- not a Python loop of per-parameter copies (96 tensors for
        the encoder, 400+ for the full model), which would cost more than the collective itself."""
        if self.world == 1:
            return
        pending = []
        for params in self.buckets:
            grads = []
            for p in params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                grads.append(p.grad)
            flat = torch.cat([g.reshape(-1) for g in grads])
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            pending.append((work, flat, grads))
        inv = 1.0 / self.world
        for work, flat, grads in pending:
            work.wait()
            flat.mul_(inv)
            views, off = [], 0
            for g in grads:
                views.append(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
            torch._foreach_copy_(grads, views)


    def average_(self, flat):
        """In-place average across ranks of gradients that are already packed in ONE flat tensor
        (``optim.FlatAdam.pack_grads``): a single large all-reduce, no pack / unpack here."""
        if self.world == 1 and not self.always:
            return flat
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat.mul_(1.0 / self.world)

    def sum_(self, flat):
        """In-place SUM across ranks of an already packed gradient tensor; the caller folds 1/world into the
        optimizer (``FlatAdam.step(flat, grad_scale=1/world)``), saving the scaling pass over the buffer."""
        if self.world > 1 or self.always:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat


def broadcast_parameters(model, src=0, group=None):
    """Make every replica start from rank ``src``'s parameters and buffers."""
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)

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
        self._flat = [None] * len(self.buckets)

    def _flat_for(self, i, params):
        if self._flat[i] is None:
            n = sum(p.numel() for p in params)
            self._flat[i] = torch.empty(n, dtype=params[0].dtype, device=params[0].device)
        return self._flat[i]

    def sync(self):
        """Average ``.grad`` of every parameter across ranks (missing grads count as zero)."""
        if self.world == 1:
            return
        pending = []
        for i, params in enumerate(self.buckets):
            flat = self._flat_for(i, params)
            off = 0
            for p in params:
                n = p.numel()
                if p.grad is None:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
                off += n
            pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True),
                            flat, params))
        inv = 1.0 / self.world
        for work, flat, params in pending:
            work.wait()
            off = 0
            for p in params:
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = (g * inv).clone()
                else:
                    torch.mul(g, inv, out=p.grad)
                off += n


def broadcast_parameters(model, src=0, group=None):
    """Make every replica start from rank ``src``'s parameters and buffers."""
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)

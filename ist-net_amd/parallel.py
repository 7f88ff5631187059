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


class OverlappedFlatReducer:
    """Sum all-reduce of ``FlatAdam.flat_grad`` in buckets, each issued AS SOON AS ITS SLICE IS FINAL, during backward.

    The reference exchanges gradients inside ``nn.DataParallel``'s backward (train.py:98-99, reduce-add onto GPU 0).
    Here the gradients of all parameters live in one flat buffer (``optim.FlatAdam.flat_grad``, written in place by the
    backward kernels); backward produces them roughly in reverse parameter order, so the buffer is cut into contiguous
    buckets of ~``bucket_bytes`` from its END, and a ``post_accumulate_grad`` hook per parameter counts a bucket down.
    When the last parameter of a bucket has its gradient, a side stream waits for the producers -- the stream backward
    runs on AND the deferred weight-gradient streams of ``fused_mlp`` (the kernels that write those slots are enqueued
    before autograd stores ``.grad``) -- and issues ONE asynchronous all-reduce of that slice.  The full IST-Net has
    107 MB of gradients (4 buckets of ~25 MB): the first three exchanges run under the rest of backward instead of
    after it.  xGMI rings are per-link bound, so few large messages: the default bucket is 25 MB.

        red = OverlappedFlatReducer(opt)           # once; opt = FlatAdam
        opt.zero_grad(set_to_none=True); loss.backward()
        opt.step(red.finish(), grad_scale=1.0 / world)

    Captured steps.  Inside a HIP-graph capture no collective is issued from the hooks; they record which parameters
    the captured backward produces a gradient for and WHERE (the static tensor autograd stored in ``.grad`` during the
    capture -- the flat slot itself for the in-place kernels, a graph-private tensor for e.g. MIOpen's gradients):

        with torch.cuda.graph(g): fwd_bwd()
        token = red.end_capture()                  # one token per captured graph
        ...
        g.replay(); opt.step(red.finish(captured=token), grad_scale=1.0 / world)

    ``finish(captured=token)`` then issues the buckets back to back -- bucketed, not overlapped with backward -- reading
    every gradient from the tensor recorded in the token.  Do NOT call ``zero_grad`` between replays: a replay writes
    its static tensors whatever ``.grad`` says, and validity is tracked here, not through ``p.grad is None``.

    Gradient accumulation (several backward passes before ``finish()``) cannot be overlapped: a bucket that already
    left was summed over ranks and would miss the second contribution.  A hook that fires for a parameter whose bucket
    has been launched raises; call ``finish()`` after every backward pass, or construct with ``overlap=False`` to defer
    every launch to ``finish()``."""

    def __init__(self, opt, world_size=None, bucket_bytes=25 << 20, group=None, always=False, overlap=True,
                 capture_collectives=False):
        self.opt, self.group = opt, group
        self.overlap = overlap     # False: hooks only mark, every bucket is launched by finish() (gradient accumulation)
        # True: a backward pass that is being CAPTURED issues its buckets from the hooks like an eager one -- the
        # collectives become nodes of the HIP graph, forked onto the communication stream as soon as their slice is
        # final, and ``finish_captured()`` joins them before the (captured) optimizer launch.  A replay then runs
        # backward, the overlapped exchange and Adam with one host call (RCCL collectives are capturable; the capture
        # must use capture_error_mode="thread_local": ProcessGroupNCCL's watchdog polls events from another thread).
        self.capture_collectives = capture_collectives
        self.world = world_size if world_size is not None else dist.get_world_size(group)
        self.always = always       # True: issue the collectives even with one rank (single-GPU dry run of the RCCL path)
        esz = opt.flat_grad.element_size()
        self.buckets = []          # [lo, hi, parameter indices], from the end of the buffer
        cur, hi = [], opt.flat_grad.numel()
        order = list(getattr(opt, "layout", range(len(opt.params))))      # parameter indices in memory order
        for k in range(len(order) - 1, -1, -1):
            i = order[k]
            cur.append(i)
            lo = opt.offsets[i]
            if (hi - lo) * esz >= bucket_bytes or k == 0:
                self.buckets.append((lo, hi, cur))
                cur, hi = [], lo
        self.bucket_of = {}
        for b, (_, _, idxs) in enumerate(self.buckets):
            for i in idxs:
                self.bucket_of[i] = b
        self.comm = torch.cuda.Stream(device=opt.flat_grad.device) if opt.flat_grad.is_cuda else None
        self._capturing = {}
        self._cap_left = [len(idxs) for _, _, idxs in self.buckets]
        self._cap_seen = [False] * len(opt.params)
        self._cap_launched = [False] * len(self.buckets)
        self._reset()
        self.issued_in_backward = 0          # statistics of the last step (tests, logging)
        for i, p in enumerate(opt.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _active(self):
        return self.world > 1 or self.always

    def _reset(self):
        self.left = [len(idxs) for _, _, idxs in self.buckets]
        self.seen = [False] * len(self.opt.params)
        self.works = [None] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.source = {}           # parameter index -> tensor holding its gradient for THIS step (captured steps)

    def _make_hook(self, i):
        def hook(param):
            b = self.bucket_of[i]
            if self._in_capture(param):
                # captured backward: remember the static tensor this graph writes the gradient to
                self._capturing[i] = param.grad
                if self.capture_collectives and self._active() and not self._cap_seen[i]:
                    self._cap_seen[i] = True
                    self._cap_left[b] -= 1
                    if self._cap_left[b] == 0:          # the slice is final: its all-reduce becomes a graph node here
                        self._launch(b, source=self._capturing)
                        self._cap_launched[b] = True
                return
            if self.seen[i]:
                if self.launched[b]:
                    raise RuntimeError(
                        "OverlappedFlatReducer: a second backward pass produced a gradient for a parameter whose bucket "
                        "was already all-reduced; call finish() after every backward pass or use overlap=False")
                return             # accumulation before the bucket left: .grad holds the sum when the bucket is launched
            self.seen[i] = True
            self.left[b] -= 1
            if self.left[b] == 0 and self._active() and self.overlap:
                self._launch(b)
                self.issued_in_backward += 1
        return hook

    _capturing = None

    @staticmethod
    def _in_capture(param):
        """Is this hook running inside a HIP-graph capture?  (A method so that the CPU tests can stand in for one.)"""
        return param.is_cuda and torch.cuda.is_current_stream_capturing()

    def begin_capture(self):
        """Start recording a captured backward (optional without ``capture_collectives``; ``end_capture`` also works
        without it)."""
        self._capturing = {}
        self._cap_left = [len(idxs) for _, _, idxs in self.buckets]
        self._cap_seen = [False] * len(self.opt.params)
        self._cap_launched = [False] * len(self.buckets)

    def finish_captured(self):
        """``capture_collectives``: call INSIDE the capture, after the captured backward and before the captured
        optimizer launch.  Issues the buckets backward did not complete (parameters without a gradient in this graph: their
        slots are zeroed), joins the communication stream into the capturing stream and returns the flat gradient buffer,
        which holds the sum over ranks when the graph is replayed.  Nothing is left to do after a replay."""
        if not self.capture_collectives:
            raise RuntimeError("finish_captured() needs OverlappedFlatReducer(..., capture_collectives=True)")
        opt = self.opt
        if self._active():
            for b in range(len(self.buckets)):
                if not self._cap_launched[b]:
                    self._launch(b, source=self._capturing)
                    self._cap_launched[b] = True
            if self.comm is not None:
                torch.cuda.current_stream(opt.flat_grad.device).wait_stream(self.comm)
            else:                               # CPU tensors (gloo tests): no streams, wait on the handles
                for w in self.works:
                    if w is not None:
                        w.wait()
        else:                                   # one rank, no collective: pack from the graph's static tensors
            for i, p in enumerate(opt.params):
                slot, g = p._istnet_grad_slot, self._capturing.get(i)
                if g is None:
                    slot.zero_()
                elif g.data_ptr() != slot.data_ptr() or not g.is_contiguous():
                    opt._view(opt.flat_grad, i).copy_(g)
        # Work handles of captured collectives are not waited on (the join above is the edge), and the eager bookkeeping
        # _launch() touched during the capture (launched / works) must not leak into the next eager backward: a bucket
        # left marked "launched" would be skipped by finish() when not all of its hooks fire (unused parameters)
        self._reset()
        return opt.flat_grad

    def end_capture(self):
        """Token of the graph captured since the last ``end_capture`` / ``begin_capture``: parameter index -> the
        static tensor that graph's backward writes the gradient to.  Pass it to ``finish(captured=...)``."""
        token, self._capturing = (self._capturing or {}), {}
        return token

    def _grad_of(self, i):
        """The tensor holding parameter i's gradient of the current step, or None if the step produced none."""
        if i in self.source:
            return self.source[i]
        return self.opt.params[i].grad if self.seen[i] else None

    def _launch(self, b, source=None):
        lo, hi, idxs = self.buckets[b]
        opt = self.opt
        for i in idxs:                     # gradients that did not arrive in place (or not at all) go into their slots
            slot, g = opt.params[i]._istnet_grad_slot, (source.get(i) if source is not None else self._grad_of(i))
            if g is None:
                slot.zero_()
            elif g.data_ptr() != slot.data_ptr() or not g.is_contiguous():
                opt._view(opt.flat_grad, i).copy_(g)      # the slot in the parameter's own (possibly channels-last) layout
        piece = opt.flat_grad[lo:hi]
        if self.comm is None:              # CPU tensors (gloo tests): no streams
            self.works[b] = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            dev = piece.device
            self.comm.wait_stream(torch.cuda.current_stream(dev))
            from .pointnet2 import fused_mlp
            for wstream in fused_mlp.deferred_streams(dev):
                self.comm.wait_stream(wstream)
            with torch.cuda.stream(self.comm):
                self.works[b] = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.launched[b] = True

    def finish(self, captured=None):
        """Issue what backward did not (unused parameters, a captured step), wait for every bucket on the current
        stream and return the summed flat gradient (``FlatAdam.step(flat, grad_scale=1/world)`` divides).
        ``captured``: the ``end_capture()`` token of the graph that was just replayed."""
        opt = self.opt
        if captured is not None:
            self.source = captured
        if self._active():
            for b in range(len(self.buckets)):
                if not self.launched[b]:
                    self._launch(b)
            for w in self.works:
                if w is not None:
                    w.wait()
            if self.comm is not None:
                torch.cuda.current_stream(opt.flat_grad.device).wait_stream(self.comm)
            out = opt.flat_grad
        elif captured is not None:         # one rank, captured step: pack from the graph's static tensors
            for i, p in enumerate(opt.params):
                slot, g = p._istnet_grad_slot, self._grad_of(i)
                if g is None:
                    slot.zero_()
                elif g.data_ptr() != slot.data_ptr() or not g.is_contiguous():
                    opt._view(opt.flat_grad, i).copy_(g)
            out = opt.flat_grad
        else:
            out = opt.pack_grads()
        self._reset()
        return out


def broadcast_parameters(model, src=0, group=None):
    """Make every replica start from rank ``src``'s parameters and buffers."""
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)

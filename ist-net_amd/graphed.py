"""Transparent HIP-graph segments for an eager caller.

The reference's training loop (utils/solver.py:88-99) is eager: ``zero_grad`` -> ``model(batch)`` -> ``loss.backward()`` ->
``optimizer.step()``.  The point encoder is ~230 kernel launches of 2-30 us each per step, so an eager caller is bound by
the host issuing them (5.8-7.6 ms per B=32 step against 2.9 ms of device time, profiles/r05_eager_baseline.txt).  On this
hardware the launch sequence of a fixed-shape step belongs in a HIP graph: ``AutoGraph`` gives an ``nn.Module`` that, for
the caller's unchanged eager loop, replays one captured graph for its forward and one for its backward.

How it works (the recipe of ``torch.cuda.make_graphed_callables``, applied lazily and per input shape so that the caller
does not have to ask for it):

* the first ``WARMUP_CALLS`` calls with a new key run the module's plain path (allocator / stream pools / library handles
  warm up, and a caller that only ever makes a few calls never pays for a capture);
* the next call captures the forward into a HIP graph on a static copy of the input, then the backward
  (``torch.autograd.grad`` of the static output w.r.t. the trainable parameters) into a second graph sharing its memory pool;
* from then on ``forward`` = copy the input into the static buffer + one graph launch, ``backward`` = copy the incoming
  gradient + one graph launch; parameter gradients come back to autograd as the static buffers the backward kernels wrote
  (with ``optim.FlatAdam`` attached those are the optimizer's flat gradient slots, so nothing is copied at all).

The kernels, their order and their arithmetic are the plain path's: results are bit-identical (tests/test_autograph_gpu.py).

The plain path is taken -- silently, it is always correct -- whenever replaying would not be: inside somebody else's stream
capture, with autograd disabled, for inputs that require grad, while the previous graphed forward's output is still alive
and has not been back-propagated (two forwards before one backward: siamese use, gradient accumulation over micro-batches
held at once), when a trainable parameter already holds a ``.grad`` (accumulation: the captured kernels overwrite their
destination), and when a module in the tree carries forward / backward hooks.  The key of a captured entry holds
everything the capture baked in: input shape / dtype / device, every module's ``training`` flag, the addresses of all
parameters and buffers, which parameters require grad, the presence of optimizer gradient slots, and the library's switch state.

Unlike ``make_graphed_callables`` the returned tensor is the caller's own (a copy of the static output buffer), so holding
outputs across steps is safe.  The one cost a caller can see is memory: an entry keeps its activations' pool (1.3 GB for the
B=32 encoder) until ``graphed.reset(module)`` or the module dies.

``InferenceGraph`` (below) is the forward-only counterpart for eval-mode, ``torch.no_grad()`` callers of the full model.

``ISTNET_AUTO_GRAPH=0`` (or ``graphed.ENABLED = False``) turns the whole mechanism off.
"""
import os
import warnings
import weakref

import torch

ENABLED = os.environ.get("ISTNET_AUTO_GRAPH", "1") != "0"
WARMUP_CALLS = 2
MAX_ENTRIES = 4          # captured shapes kept per module (least recently used goes first)
import collections
WHY = collections.Counter()      # AutoGraph: why a call took the plain path (diagnostics; tests print it when a capture is missing)
STATS = {"captures": 0, "replays": 0, "plain": 0, "failed": 0,                       # AutoGraph (training)
         "infer_captures": 0, "infer_replays": 0, "infer_plain": 0, "infer_failed": 0}    # InferenceGraph
# other threads of the caller (a DataLoader's pin-memory thread, a logger) keep making HIP calls while this thread captures:
# only this thread's calls are checked against the capture
_CAPTURE_MODE = "thread_local"


class _Entry:
    __slots__ = ("calls", "failed", "fwd", "bwd", "static_in", "static_out", "static_gout", "static_grads", "params",
                 "out_ref", "pending", "pool", "stamp")

    def __init__(self):
        self.calls, self.failed, self.fwd, self.bwd = 0, False, None, None
        self.out_ref, self.pending, self.stamp = None, False, 0


class _Replay(torch.autograd.Function):
    """forward = input copy + forward-graph launch; backward = gradient copy + backward-graph launch."""

    @staticmethod
    def forward(ctx, entry, x, *params):
        ctx.entry = entry
        with torch.cuda.device(x.device):       # a graph launches on the CURRENT device's current stream
            if x.data_ptr() != entry.static_in.data_ptr():
                entry.static_in.copy_(x)
            entry.fwd.replay()
            entry.pending = True
            # a copy (16 MB at B=32, ~10 us): the caller owns its output like on the plain path, whatever it keeps across steps
            return entry.static_out.clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        entry = ctx.entry
        with torch.cuda.device(entry.static_gout.device):
            if gout.data_ptr() != entry.static_gout.data_ptr():
                entry.static_gout.copy_(gout)
            entry.bwd.replay()
        entry.pending = False
        # fresh tensor objects over the static buffers: AccumulateGrad then stores them instead of cloning
        return (None, None) + tuple(None if g is None else g.detach() for g in entry.static_grads)


class AutoGraph:
    """Per-module cache of captured (forward, backward) graph pairs; see the module docstring.

    ``plain`` is the module's ordinary forward as a callable of ONE tensor returning ONE tensor."""

    def __init__(self, plain, switch_state=None):
        self._plain = plain                  # plain(module, x) -> tensor
        self.switch_state = switch_state or (lambda: ())
        self.entries = {}
        self._stamp = 0
        self.module = None                   # weak reference, set by for_module()
        self._mods = None                    # the module tree, flattened once

    def plain(self, x):
        return self._plain(self.module(), x)

    # -- what a capture bakes in -------------------------------------------------------------------------------------
    def _key(self, module, x):
        # the module TREE is walked once (nn.Module's generators cost ~0.7 ms per call for the encoder's 168 modules); what
        # the modules hold -- parameters, buffers, flags, hooks -- is read afresh every call.  A submodule added or replaced
        # after the first call needs graphed.reset(module).
        mods = self._mods
        if mods is None:
            mods = self._mods = list(module.modules())
        flags, addrs, req = [], [], []
        for m in mods:
            flags.append(m.training)
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
                WHY[f"module hook on {type(m).__name__}"] += 1
                return None
            for p in m._parameters.values():
                if p is None:
                    continue
                addrs.append(p.data_ptr())
                req.append(p.requires_grad)
                if p.requires_grad and p.grad is not None:
                    WHY["a parameter already holds a .grad (accumulation)"] += 1
                    return None              # accumulation into an existing .grad: the captured kernels overwrite
                slot = p.__dict__.get("_istnet_grad_slot")
                addrs.append(0 if slot is None else slot.data_ptr())
            for b in m._buffers.values():
                if b is not None:
                    addrs.append(b.data_ptr())
        return (tuple(x.shape), x.dtype, x.device, tuple(flags), tuple(addrs), tuple(req), self.switch_state())

    def __call__(self, x):
        module = self.module()
        if not (ENABLED and x.is_cuda and torch.is_grad_enabled() and not x.requires_grad and x.is_contiguous()
                and not torch.cuda.is_current_stream_capturing()):
            WHY["not eligible (disabled / no grad mode / input requires grad / capturing)"] += 1
            return self.plain(x)
        key = self._key(module, x)
        if key is None or not any(key[5]):
            STATS["plain"] += 1
            WHY["hooks, an existing .grad, or nothing to differentiate"] += 1
            return self.plain(x)
        entry = self.entries.get(key)
        if entry is None:
            if len(self.entries) >= MAX_ENTRIES:          # shapes come and go (last partial batch): keep the recent ones
                oldest = min(self.entries, key=lambda k: self.entries[k].stamp)
                del self.entries[oldest]
            entry = self.entries[key] = _Entry()
        self._stamp += 1
        entry.stamp = self._stamp
        entry.calls += 1
        if entry.failed or entry.calls <= WARMUP_CALLS:
            STATS["plain"] += 1
            WHY["capture failed earlier" if entry.failed else "warm-up call"] += 1
            return self.plain(x)
        if entry.fwd is None:
            try:
                self._capture(entry, module, x)
            except Exception as exc:            # capture refused (unsupported op, allocator state ...): plain path for good
                entry.failed = True
                entry.fwd = entry.bwd = None
                STATS["failed"] += 1
                torch.cuda.synchronize(x.device)
                warnings.warn(f"istnet_amd.graphed: HIP-graph capture of {type(module).__name__} failed "
                              f"({type(exc).__name__}: {exc}); this shape keeps running launch by launch", RuntimeWarning)
                return self.plain(x)
        if entry.pending and entry.out_ref is not None and entry.out_ref() is not None:
            # the previous graphed forward has not been back-propagated and its output is still referenced: a replay
            # would overwrite the activations that backward needs
            STATS["plain"] += 1
            WHY["previous graphed output still alive and not back-propagated"] += 1
            return self.plain(x)
        params = entry.params
        out = _Replay.apply(entry, x, *params)
        entry.out_ref = weakref.ref(out)
        STATS["replays"] += 1
        return out

    def _capture(self, entry, module, x):
        dev = x.device
        params = [p for p in module.parameters() if p.requires_grad]
        # The capture differentiates w.r.t. fresh leaf ALIASES of the parameters (same storage), swapped into the modules
        # for the duration of the forward capture.  The engine synchronises a gradient with the stream its leaf's
        # AccumulateGrad node was created on; a parameter whose node is kept alive by an autograd graph the caller still
        # holds (last step's loss / output) has the caller's stream there -- usually the default stream -- and that sync
        # pulls the default stream into the capture, which then cannot end (hipStreamEndCapture dies on this stack).  An
        # alias gets its node inside the capture, on the capture stream.
        alias, swaps = {}, []
        for m in module.modules():
            for name, p in list(m._parameters.items()):
                if p is None or not p.requires_grad:
                    continue
                a = alias.get(id(p))
                if a is None:
                    a = alias[id(p)] = p.detach().requires_grad_(True)
                    slot = getattr(p, "_istnet_grad_slot", None)
                    if slot is not None:
                        a._istnet_grad_slot = slot          # optim.FlatAdam: the backward kernels write the flat gradient
                swaps.append((m, name, p))
                m._parameters[name] = a
        try:
            with torch.cuda.device(dev):
                torch.cuda.synchronize(dev)
                pool = torch.cuda.graph_pool_handle()
                static_in = x.detach().clone()
                fwd = torch.cuda.CUDAGraph()
                with torch.cuda.graph(fwd, pool=pool, capture_error_mode=_CAPTURE_MODE):
                    static_out = self.plain(static_in)
                if not (isinstance(static_out, torch.Tensor) and static_out.requires_grad):
                    raise RuntimeError("the module's output is not a differentiable tensor")
                static_gout = torch.empty_like(static_out)
                bwd = torch.cuda.CUDAGraph()
                with torch.cuda.graph(bwd, pool=pool, capture_error_mode=_CAPTURE_MODE):
                    grads = torch.autograd.grad((static_out,), [alias[id(p)] for p in params], (static_gout,),
                                                allow_unused=True)
                torch.cuda.synchronize(dev)
        finally:
            for m, name, p in swaps:
                m._parameters[name] = p
        entry.fwd, entry.bwd, entry.pool = fwd, bwd, pool
        entry.static_in, entry.static_out, entry.static_gout = static_in, static_out.detach(), static_gout
        entry.static_grads, entry.params = list(grads), params
        STATS["captures"] += 1

    def reset(self):
        self.entries.clear()
        self._mods = None


class InferenceGraph:
    """Forward-only graphs for an eval-mode, ``torch.no_grad()`` caller whose module maps a dict of tensors to a dict of tensors.

    The reference's test loop (test.py / utils/solver.py:199-262) feeds ONE IMAGE per step: the batch is that image's
    instances, usually 1-8.  At that size the forward pass of the full model is ~600 launches of a few microseconds each and
    the step is the host issuing them: 5.0 ms per image whatever the batch, against 1.7-2.4 ms of device time
    (profiles/r05_infer_small_batch.txt).  Same recipe as ``AutoGraph``: per input signature (every tensor's key, shape, dtype
    and contiguity) two plain warm-up calls, then the whole forward pass captured once on static copies of the inputs and
    replayed -- copy the inputs in, one graph launch, hand back copies of the outputs.  A test set's images have a handful of
    different instance counts; ``INFER_MAX_ENTRIES`` signatures are kept (least recently used goes first).

    The key also holds what the capture baked in: the addresses of every parameter and buffer (``load_state_dict`` copies in
    place and is picked up by the next replay; ``.to()`` / ``.half()`` move the storage and start a new entry), every module's
    ``training`` flag and the library's switch state.  The parameter / buffer / module LISTS are flattened once: a caller that
    replaces Parameter objects or submodules after the first call (pruning, weight tying) calls ``graphed.reset(module)``.
    Modules carrying hooks, inputs that require grad, a surrounding stream capture and ``ISTNET_AUTO_GRAPH=0`` keep the
    plain path; a capture that fails (an extractor with host-side control flow) warns once and keeps it for that signature."""

    def __init__(self, plain, switch_state=None):
        self._plain = plain                  # plain(module, inputs) -> dict of tensors
        self.switch_state = switch_state or (lambda: ())
        self.entries = {}
        self._stamp = 0
        self.module = None
        self._mods = self._tensors = None
        self._hooked = False

    def plain(self, inputs):
        return self._plain(self.module(), inputs)

    def _fingerprint(self, module):
        if self._tensors is None:
            self._mods = list(module.modules())
            self._hooked = any(m._forward_hooks or m._forward_pre_hooks for m in self._mods)
            self._tensors = list(module.parameters()) + list(module.buffers())
        return (tuple(t.data_ptr() for t in self._tensors), tuple(m.training for m in self._mods))

    def __call__(self, inputs, keys):
        first = inputs[keys[0]]
        if not (ENABLED and isinstance(first, torch.Tensor) and first.is_cuda and not torch.is_grad_enabled()
                and not torch.cuda.is_current_stream_capturing()):
            return self.plain(inputs)
        sig = []
        for k in keys:
            t = inputs[k]
            if not isinstance(t, torch.Tensor) or t.device != first.device or t.requires_grad:
                return self.plain(inputs)
            sig.append((k, tuple(t.shape), t.dtype, t.is_contiguous()))
        module = self.module()
        fp = self._fingerprint(module)
        if self._hooked or module._forward_hooks or module._forward_pre_hooks:
            STATS["infer_plain"] += 1
            return self.plain(inputs)
        key = (tuple(sig), fp, self.switch_state())
        entry = self.entries.get(key)
        if entry is None:
            if len(self.entries) >= INFER_MAX_ENTRIES:
                del self.entries[min(self.entries, key=lambda k: self.entries[k].stamp)]
            entry = self.entries[key] = _Entry()
        self._stamp += 1
        entry.stamp = self._stamp
        entry.calls += 1
        if entry.failed or entry.calls <= WARMUP_CALLS:
            STATS["infer_plain"] += 1
            return self.plain(inputs)
        dev = first.device
        if entry.fwd is None:
            try:
                with torch.cuda.device(dev):
                    torch.cuda.synchronize(dev)
                    static_in = {k: inputs[k].detach().clone() for k in keys}
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, capture_error_mode=_CAPTURE_MODE):
                        static_out = self.plain(static_in)     # ONLY the declared keys: nothing else can be baked in
                    if not (isinstance(static_out, dict) and all(isinstance(v, torch.Tensor) for v in static_out.values())):
                        raise RuntimeError("the module's output is not a dict of tensors")
                    torch.cuda.synchronize(dev)
                entry.fwd, entry.static_in, entry.static_out = graph, static_in, static_out
                STATS["infer_captures"] += 1
            except Exception as exc:
                entry.failed, entry.fwd = True, None
                STATS["infer_failed"] += 1
                torch.cuda.synchronize(dev)
                warnings.warn(f"istnet_amd.graphed: HIP-graph capture of {type(module).__name__}'s inference pass failed "
                              f"({type(exc).__name__}: {exc}); this input signature keeps running launch by launch", RuntimeWarning)
                return self.plain(inputs)
        with torch.cuda.device(dev):
            for k in keys:
                if inputs[k].data_ptr() != entry.static_in[k].data_ptr():
                    entry.static_in[k].copy_(inputs[k])
            entry.fwd.replay()
            STATS["infer_replays"] += 1
            return {k: v.clone() for k, v in entry.static_out.items()}     # the caller owns its outputs

    def reset(self):
        self.entries.clear()
        self._mods = self._tensors = None


INFER_MAX_ENTRIES = 12      # input signatures (batch sizes) kept per module

_REGISTRY = weakref.WeakKeyDictionary()      # module -> AutoGraph (kept outside the module: deepcopy / pickle stay plain)
_INFER_REGISTRY = weakref.WeakKeyDictionary()      # module -> InferenceGraph


def for_module(module, plain, switch_state=None):
    """The AutoGraph of ``module`` (created on first use).  ``plain(module, x)`` is the module's ordinary forward."""
    ag = _REGISTRY.get(module)
    if ag is None:
        ag = _REGISTRY[module] = AutoGraph(plain, switch_state)
        ag.module = weakref.ref(module)
    return ag


def for_inference(module, plain, switch_state=None):
    """The InferenceGraph of ``module`` (created on first use).  ``plain(module, inputs)`` is the module's ordinary forward."""
    ig = _INFER_REGISTRY.get(module)
    if ig is None:
        ig = _INFER_REGISTRY[module] = InferenceGraph(plain, switch_state)
        ig.module = weakref.ref(module)
    return ig


def reset(module=None):
    """Drop captured graphs (of one module and of the modules inside it, or of all): frees their memory pools."""
    inside = None if module is None else {id(m) for m in module.modules()}
    for registry in (_REGISTRY, _INFER_REGISTRY):
        for m, ag in list(registry.items()):
            if inside is None or id(m) in inside:
                ag.reset()

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

What a replay reads from HOST state each time, because the caller may change it between two steps without going through
this package: every BatchNorm's ``momentum`` (the reference's own ``BNMomentumScheduler`` -- utils/scheduler.py, built at
utils/solver.py:48-49 and stepped at :92 -- is a bare ``m.momentum = x``): the finalize kernels take the momentum from a
device slot, and ``AutoGraph`` compares every module's host value with its slot's on every call and refreshes the slots (one
pinned host-to-device copy, only when something changed) before it replays.  ``eps`` and ``track_running_stats`` are baked
into the captured launches and therefore part of the key.

Lifetime of the captured graphs: an entry is destroyed explicitly -- on the calling thread, with the device idle -- when it is
evicted (``MAX_ENTRIES`` shapes per module, least recently used first) or reset; entries whose module died are parked in a
graveyard by the module's finalizer (which may run on any thread, e.g. the autograd worker during a collection) and
destroyed by the next call or ``reset()`` on the caller's thread.  A capture logs the memory it pins
(``logging.getLogger("istnet_amd.graphed")``, INFO).

``InferenceGraph`` (below) is the forward-only counterpart for eval-mode, ``torch.no_grad()`` callers of the full model.

``ISTNET_AUTO_GRAPH=0`` (or ``graphed.ENABLED = False``) turns the whole mechanism off.
"""
import collections
import logging
import os
import warnings
import weakref

import torch
from torch import nn

ENABLED = os.environ.get("ISTNET_AUTO_GRAPH", "1") != "0"
WARMUP_CALLS = 2
# captured shapes kept per module (least recently used goes first).  An entry pins its activations' pool -- 1.3 GB for the
# B=32 encoder -- and IST_Net holds three encoders, so the default is small; a loader with more than two batch shapes in
# rotation raises it (ISTNET_AUTO_GRAPH_ENTRIES).
MAX_ENTRIES = int(os.environ.get("ISTNET_AUTO_GRAPH_ENTRIES", "2"))
PLAIN_STREAK_WARN = 10   # consecutive plain-path calls of a TRAINING module after which the caller is told why (once)
_LOG = logging.getLogger("istnet_amd.graphed")
WHY = collections.Counter()      # AutoGraph: why a call took the plain path (diagnostics; tests print it when a capture is missing)
STATS = {"captures": 0, "replays": 0, "plain": 0, "failed": 0, "momentum_syncs": 0,  # AutoGraph (training)
         "infer_captures": 0, "infer_replays": 0, "infer_plain": 0, "infer_failed": 0}    # InferenceGraph
# other threads of the caller (a DataLoader's pin-memory thread, a logger) keep making HIP calls while this thread captures:
# only this thread's calls are checked against the capture
_CAPTURE_MODE = "thread_local"


_BN_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)
_GRAVEYARD = []          # entries of dead modules, parked by their finalizers; destroyed by _drain() on a caller's thread


class _Entry:
    __slots__ = ("calls", "failed", "fwd", "bwd", "static_in", "static_out", "static_gout", "static_grads", "params",
                 "node_ref", "pending", "pool", "stamp", "gen", "bwd_gen", "baked", "__weakref__")

    def __init__(self):
        self.calls, self.failed, self.fwd, self.bwd = 0, False, None, None
        self.node_ref, self.pending, self.stamp = None, False, 0
        self.gen, self.bwd_gen = 0, -1       # replays so far; the replay whose activations were last back-propagated
        self.baked = None                    # momenta the capture baked in BY VALUE (a torch-composition fallback ran in it)
        self.static_in = self.static_out = self.static_gout = self.static_grads = self.params = self.pool = None

    def destroy(self):
        """Release the graphs and their pool NOW, on this thread (the caller has synchronised the device).  An autograd
        node that still holds this entry finds it dead and says so."""
        self.fwd = self.bwd = None
        self.static_in = self.static_out = self.static_gout = self.static_grads = self.params = self.pool = None
        self.failed, self.pending = True, False


def _bury(entries):
    """Finalizer of a module with captured entries: runs wherever the collector happens to run, so it only parks them."""
    _GRAVEYARD.extend(entries.values())
    entries.clear()


def _drain():
    """Destroy parked entries on the calling thread with the device idle."""
    if not _GRAVEYARD or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        return                               # (inside somebody's capture nothing may synchronise: the next call drains)
    dead = list(_GRAVEYARD)
    del _GRAVEYARD[:len(dead)]               # a finalizer on another thread appends behind these
    if any(e.fwd is not None for e in dead) and torch.cuda.is_available():
        torch.cuda.synchronize()
    for e in dead:
        e.destroy()


class _Replay(torch.autograd.Function):
    """forward = input copy + forward-graph launch; backward = gradient copy + backward-graph launch."""

    @staticmethod
    def forward(ctx, entry, x, *params):
        ctx.entry = entry
        # the parameters are saved although the captured kernels read them through their addresses: autograd's version
        # counters then refuse forward -> optimizer.step() -> backward exactly as they do on the plain path
        # (tests/test_optim.py::test_backward_after_native_step_raises; FlatAdam.step bumps the versions)
        ctx.save_for_backward(*params)
        with torch.cuda.device(x.device):       # a graph launches on the CURRENT device's current stream
            if x.data_ptr() != entry.static_in.data_ptr():
                entry.static_in.copy_(x)
            entry.fwd.replay()
            entry.gen += 1
            ctx.gen = entry.gen
            entry.pending = True
            # a copy (16 MB at B=32, ~10 us): the caller owns its output like on the plain path, whatever it keeps across steps
            return entry.static_out.clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        entry = ctx.entry
        ctx.saved_tensors                        # version check of the parameters (raises like the plain path would)
        if entry.fwd is None:
            raise RuntimeError("istnet_amd.graphed: the captured graphs of this forward were destroyed (graphed.reset / "
                               "eviction of its shape) before its backward ran")
        if ctx.gen != entry.gen:
            raise RuntimeError("istnet_amd.graphed: a later forward of the same shape replayed over the activations this "
                               "backward needs (the autograd node outlived its output); ISTNET_AUTO_GRAPH=0 runs such a "
                               "graph launch by launch")
        if entry.bwd_gen == ctx.gen:
            raise RuntimeError("istnet_amd.graphed: second backward through a graph segment (retain_graph=True): the "
                               "captured backward recycles the memory of the saved activations; set ISTNET_AUTO_GRAPH=0 "
                               "for double backward passes")
        with torch.cuda.device(entry.static_gout.device):
            if gout.data_ptr() != entry.static_gout.data_ptr():
                entry.static_gout.copy_(gout)
            entry.bwd.replay()
        entry.pending = False
        entry.bwd_gen = ctx.gen
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
        self._mods = None                    # the module tree, flattened once (weak references: this object is the VALUE of
        #                                      a WeakKeyDictionary keyed by the module and must not keep it alive)
        self._plain_streak, self._warned = 0, False

    def plain(self, x):
        return self._plain(self.module(), x)

    def _note_plain(self, module, reason):
        """Count a plain-path call; tell a training caller once when it never leaves the plain path."""
        STATS["plain"] += 1
        WHY[reason] += 1
        self._plain_streak += 1
        if (self._plain_streak >= PLAIN_STREAK_WARN and not self._warned and module.training
                and reason not in ("warm-up call",)):
            self._warned = True
            warnings.warn(f"istnet_amd.graphed: {type(module).__name__} has run launch by launch for {self._plain_streak} "
                          f"consecutive calls ({reason}); the HIP-graph path is ~1.7x faster for an eager training loop. "
                          "Typical cause: optimizer.zero_grad(set_to_none=False) keeps .grad tensors alive -- use "
                          "set_to_none=True (the torch >= 2.0 default).  graphed.WHY has the counts.", RuntimeWarning)

    # -- what a capture bakes in -------------------------------------------------------------------------------------
    def _key(self, module, x):
        # the module TREE is walked once (nn.Module's generators cost ~0.7 ms per call for the encoder's 168 modules); what
        # the modules hold -- parameters, buffers, flags, hooks -- is read afresh every call.  A submodule added or replaced
        # after the first call needs graphed.reset(module).
        refs = self._mods
        if refs is None:
            refs = self._mods = [weakref.ref(m) for m in module.modules()]
        flags, addrs, req = [], [], []
        for r in refs:
            m = r()
            if m is None:
                return "submodule gone"
            flags.append(m.training)
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
                return f"module hook on {type(m).__name__}"
            if isinstance(m, _BN_TYPES):
                # eps and track_running_stats are arguments of the captured launches; the momentum is NOT part of the key:
                # it is read from a device slot that __call__ refreshes from the host value before every replay
                flags.append(m.eps)
                flags.append(m.track_running_stats)
            for p in m._parameters.values():
                if p is None:
                    continue
                addrs.append(p.data_ptr())
                req.append(p.requires_grad)
                if p.requires_grad and p.grad is not None:
                    return "a parameter already holds a .grad (accumulation, or zero_grad(set_to_none=False))"
                slot = p.__dict__.get("_istnet_grad_slot")
                addrs.append(0 if slot is None else slot.data_ptr())
            for b in m._buffers.values():
                if b is not None:
                    addrs.append(b.data_ptr())
        return (tuple(x.shape), x.dtype, x.device, tuple(flags), tuple(addrs), tuple(req), self.switch_state(),
                torch.is_autocast_enabled())

    def _follow_momentum(self, module):
        """Mirror every BatchNorm's host ``momentum`` into its device slot when any differs (utils/scheduler.py's
        BNMomentumScheduler and a bare ``model.apply(lambda m: setattr(m, "momentum", x))`` both land here)."""
        from .pointnet2.pytorch_utils import _MomentumSlots
        bufs = _MomentumSlots._bufs
        for r in self._mods:
            m = r()
            rec = m.__dict__.get("_istnet_mslot") if isinstance(m, _BN_TYPES) else None
            if rec is not None and m.momentum is not None and bufs[rec[0]][2][rec[1]] != float(m.momentum):
                _MomentumSlots.sync(module)
                STATS["momentum_syncs"] += 1
                return

    def _momenta(self):
        return tuple(m.momentum for m in (r() for r in self._mods) if isinstance(m, _BN_TYPES))

    def _evict(self, key):
        entry = self.entries.pop(key)
        if entry.fwd is not None:
            torch.cuda.synchronize()
        entry.destroy()

    def __call__(self, x):
        module = self.module()
        _drain()
        if not (ENABLED and x.is_cuda and torch.is_grad_enabled() and not x.requires_grad and x.is_contiguous()
                and not torch.cuda.is_current_stream_capturing()):
            WHY["not eligible (disabled / no grad mode / input requires grad / capturing)"] += 1
            return self.plain(x)
        key = self._key(module, x)
        if isinstance(key, str) or not any(key[5]):
            self._note_plain(module, key if isinstance(key, str) else "nothing to differentiate")
            return self.plain(x)
        entry = self.entries.get(key)
        if entry is None:
            while len(self.entries) >= max(MAX_ENTRIES, 1):   # shapes come and go (last partial batch): keep the recent ones
                self._evict(min(self.entries, key=lambda k: self.entries[k].stamp))
            entry = self.entries[key] = _Entry()
        self._stamp += 1
        entry.stamp = self._stamp
        entry.calls += 1
        if entry.failed or entry.calls <= WARMUP_CALLS:
            self._note_plain(module, "capture failed earlier" if entry.failed else "warm-up call")
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
        if entry.pending and entry.node_ref is not None and entry.node_ref() is not None:
            # the previous graphed forward has not been back-propagated and its autograd NODE is still reachable (the
            # output tensor itself may be gone: mean / sum / cat / slicing do not keep their input): a replay would
            # overwrite the activations that node's backward needs
            self._note_plain(module, "previous graphed forward still awaits its backward")
            return self.plain(x)
        if entry.baked is not None and entry.baked != self._momenta():
            # a torch-composition fallback inside the capture took bn.momentum by value and the caller has changed it since:
            # this entry is stale for good -- drop it (the shape warms up and captures again with today's values)
            self._evict(key)
            self._note_plain(module, "momentum changed under a capture that baked it in (torch fallback inside)")
            return self.plain(x)
        self._follow_momentum(module)
        out = _Replay.apply(entry, x, *entry.params)
        entry.node_ref = weakref.ref(out.grad_fn)
        STATS["replays"] += 1
        self._plain_streak = 0
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
        # the slots must hold the host momenta BEFORE the capture (a fill recorded in the graph would put today's value
        # back on every replay; pytorch_utils._MomentumSlots.ptr refuses to do that)
        from .pointnet2 import fused_mlp
        from .pointnet2.pytorch_utils import sync_bn_momentum
        sync_bn_momentum(module)
        fallbacks = sum(fused_mlp.FALLBACKS.values())
        try:
            with torch.cuda.device(dev):
                torch.cuda.synchronize(dev)
                before = torch.cuda.memory_reserved(dev)
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
                pinned = torch.cuda.memory_reserved(dev) - before
        finally:
            for m, name, p in swaps:
                m._parameters[name] = p
        entry.fwd, entry.bwd, entry.pool = fwd, bwd, pool
        entry.static_in, entry.static_out, entry.static_gout = static_in, static_out.detach(), static_gout
        entry.static_grads, entry.params = list(grads), params
        # a shape the fused kernels do not cover ran torch's BatchNorm inside the capture: that launch took the momentum by
        # value, so the entry is only valid while the host momenta stay what they were
        entry.baked = self._momenta() if sum(fused_mlp.FALLBACKS.values()) != fallbacks else None
        STATS["captures"] += 1
        _LOG.info("captured %s for input %s: forward + backward HIP graphs, ~%.0f MB of device memory held until "
                  "graphed.reset(module), eviction (%d shapes kept per module) or the module's death",
                  type(module).__name__, tuple(x.shape), pinned / 2**20, MAX_ENTRIES)

    def reset(self):
        """Destroy every captured entry now (calling thread, device idle)."""
        if any(e.fwd is not None for e in self.entries.values()) and torch.cuda.is_available():
            torch.cuda.synchronize()
        for e in self.entries.values():
            e.destroy()
        self.entries.clear()
        self._mods = None
        self._plain_streak, self._warned = 0, False


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

    def plain(self, inputs):
        return self._plain(self.module(), inputs)

    def _fingerprint(self, module):
        for attempt in (0, 1):
            if self._tensors is None:
                self._mods = [weakref.ref(m) for m in module.modules()]
                self._tensors = [weakref.ref(t) for t in list(module.parameters()) + list(module.buffers())]
            mods = [r() for r in self._mods]
            tens = [r() for r in self._tensors]
            if all(m is not None for m in mods) and all(t is not None for t in tens):
                break
            self._tensors = None             # a buffer object was replaced (``.to()`` / ``.half()``): flatten again, once
            if attempt:
                return None
        # hooks are read afresh on every call (a profiler or feature extractor may register one at any time): a replay
        # would silently stop firing them
        if any(m._forward_hooks or m._forward_pre_hooks for m in mods):
            return None
        return (tuple(t.data_ptr() for t in tens), tuple(m.training for m in mods))

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
        _drain()
        fp = self._fingerprint(module)
        if fp is None:                       # a module carries a hook (or the tree changed): launch by launch
            STATS["infer_plain"] += 1
            return self.plain(inputs)
        key = (tuple(sig), fp, self.switch_state())
        entry = self.entries.get(key)
        if entry is None:
            while len(self.entries) >= max(INFER_MAX_ENTRIES, 1):
                old = self.entries.pop(min(self.entries, key=lambda k: self.entries[k].stamp))
                if old.fwd is not None:
                    torch.cuda.synchronize()
                old.destroy()                # on this thread, device idle -- not whenever the collector gets to it
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
        if any(e.fwd is not None for e in self.entries.values()) and torch.cuda.is_available():
            torch.cuda.synchronize()
        for e in self.entries.values():
            e.destroy()
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
        weakref.finalize(module, _bury, ag.entries)     # the graphs of a dead module are parked, never freed by the collector
    return ag


def for_inference(module, plain, switch_state=None):
    """The InferenceGraph of ``module`` (created on first use).  ``plain(module, inputs)`` is the module's ordinary forward."""
    ig = _INFER_REGISTRY.get(module)
    if ig is None:
        ig = _INFER_REGISTRY[module] = InferenceGraph(plain, switch_state)
        ig.module = weakref.ref(module)
        weakref.finalize(module, _bury, ig.entries)
    return ig


def reset(module=None):
    """Drop captured graphs (of one module and of the modules inside it, or of all): frees their memory pools."""
    _drain()
    inside = None if module is None else {id(m) for m in module.modules()}
    for registry in (_REGISTRY, _INFER_REGISTRY):
        for m, ag in list(registry.items()):
            if inside is None or id(m) in inside:
                ag.reset()

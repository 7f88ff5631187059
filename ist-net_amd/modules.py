"""Point-cloud encoder of IST-Net.

``PointNet2MSG`` mirrors model/modules.py:244-327: four 2-scale MSG set-abstraction levels
(npoint 512/256/128/64, nsample [16, 32]) followed by four feature-propagation levels, returning
per-point features (B, 128, N).  Child names (``SA_modules``, ``FP_modules``) and layer widths are
the reference's so its state dicts load unchanged.
"""
import os

import weakref

import torch
import torch.nn as nn

from . import _native, graphed
from .pointnet2 import fused_mlp, pointnet2_utils
from .pointnet2.fused_mlp import defer_bn_counters
from .pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG

# Geometry pre-pass: FPS, centroid gather, ball queries and three_nn of ALL levels depend on xyz only.
# They are tiny-grid, latency-bound kernels (FPS is 956 dependent rounds per cloud), so they are issued
# on a second HIP stream at the start of the forward and overlap the MFMA stacks of the earlier levels;
# the main stream waits on one event per level.
USE_GEOMETRY_STREAM = os.environ.get("ISTNET_GEOMETRY_STREAM", "1") != "0"
# Level l+1 samples from level l's picks in pick order, so its picks are the prefix 0..m-1 of its input whenever no
# arg-max tie occurred in the parent's first m rounds: the parent run reports its first tied round and the children
# skip their scan (include/istnet_pn2.h, istnet_pn2_fps_gather_chain).  Bit-identical to sampling every level.
# Largest cloud whose sampling run keeps the tie bookkeeping its child needs.  Measured (profiles/r04_fps_chain.txt): with the
# encoder's own sizes the full chain wins at n = 2048 too (B = 64: 466 us every level scanned, 421 us chained from level 2
# on, 367 us chained throughout; config 5 end to end 18.46 -> 18.40 ms), so the default keeps every level tracked.
FPS_CHAIN_MAX_TRACKED_N = 4096
_GEOMETRY_STREAMS = {}


def _geometry_stream(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _GEOMETRY_STREAMS:
        _GEOMETRY_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _GEOMETRY_STREAMS[key]

# (npoint, per-scale output width) of the four SA levels  [ref :249-297]
_SA_LEVELS = ((512, (16, 16, 32)), (256, (32, 32, 64)), (128, (64, 64, 128)), (64, (128, 128, 256)))
_NSAMPLES = (16, 32)


class GeometrySlot:
    """Persistent device buffers for the coordinate-only results of ONE batch (FPS picks, ball-query indices,
    three_nn indices / weights / inverse lists), written by ``PointNet2MSG.prefetch_geometry`` and read by
    ``forward(..., geometry=slot)``.  Two slots ping-pong in a pipelined training loop."""

    def __init__(self):
        self.sa = None       # per level: (new_xyz, [idx per scale], [inverse lists per scale] or None,
                             #             [compact-column tables per scale] or None)
        self.fp = None       # per FP level: (idx, weight, csr or None)
        self.event = None    # recorded on the geometry stream after the last write (eager mode)
        self.shape = None
        self.flat = None     # dtype -> flat storage the tensors above are views of
        self.plan = None     # _ext.OutputPlan: which output allocation of a prefetch pass is which view (prefetch_geometry)
        self.packed = True   # the views lie back to back in ``flat`` (no alignment padding between them)
        self.after_sa = None  # callable run ONCE by forward(geometry=self) between the set-abstraction and the feature-
                              # propagation levels (bench.py: the second part of the next batch's split prefetch)
        self.grad_mode = None

    def tensors(self):
        out = []
        for new_xyz, idxs, csrs, comps in self.sa:
            out += [new_xyz, *idxs]
            for csr in (csrs or ()):
                out += list(csr)
            for comp in (comps or ()):
                out += comp.tensors()
        for idx, weight, csr in self.fp:
            out += [idx, weight, *(csr if csr is not None else ())]
        return out


def _switch_state():
    """Everything process-global that a captured forward / backward bakes in (graphed.AutoGraph keys on it)."""
    return (fused_mlp.switch_state(), USE_GEOMETRY_STREAM, FPS_CHAIN_MAX_TRACKED_N,
            id(pointnet2_utils._ext))


class PointNet2MSG(nn.Module):
    def __init__(self, radii_list, use_xyz=True):
        super().__init__()
        self.SA_modules = nn.ModuleList()
        widths = [0]  # feature width entering each level (level 0 has xyz only)
        for (npoint, hidden), radii in zip(_SA_LEVELS, radii_list):
            c_in = widths[-1]
            self.SA_modules.append(PointnetSAModuleMSG(
                npoint=npoint, radii=list(radii), nsamples=list(_NSAMPLES),
                mlps=[[c_in, *hidden] for _ in _NSAMPLES], use_xyz=use_xyz, bn=True))
            widths.append(hidden[-1] * len(_NSAMPLES))
        c0, c1, c2, c3 = widths[1:]
        # FP_modules[i] refines level i from level i+1  [ref :299-304]
        self.FP_modules = nn.ModuleList([
            PointnetFPModule(mlp=[256, 128, 128], bn=True),
            PointnetFPModule(mlp=[256 + c0, 256, 256], bn=True),
            PointnetFPModule(mlp=[512 + c1, 256, 256], bn=True),
            PointnetFPModule(mlp=[c3 + c2, 512, 512], bn=True),
        ])

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    # ---- compact columns by what the data looks like (round 6) ----
    # A level in fused_mlp.COMPACT_LEVELS evaluates its scales on compact columns (distinct neighbours + one weighted
    # representative of the padded repeats).  That pays when balls are under-full -- shell clouds: 17 % / 33 % of the padded
    # columns are distinct at level 1, cube 8 % / 13 % -- and LOSES when they are not: on a dense cloud (every ball over-full)
    # the compact kernels process every column at a lower rate, 2.99 against 2.62 ms per step (profiles/r06_compact_by_input.txt).
    # Policy "auto": the first COMPACT_PROBES passes of a model (the eager warm-up calls that precede any capture) read the
    # valid-column count back -- one small device-to-host copy each, the only synchronisation this module ever does -- and the
    # level keeps its compact columns only if they fill at most COMPACT_MAX_FILL of the padded capacity.  The decision is per
    # model and per level, taken once (a later change of the data's character changes the speed, never the results);
    # ISTNET_COMPACT_POLICY=on / off forces it.
    def _compact_wanted(self, level):
        policy = fused_mlp.COMPACT_POLICY
        if policy != "auto":
            return policy != "off"
        return self.__dict__.setdefault("_compact_off", {}).get(level) is not True

    def _compact_probe(self, level, comps):
        if fused_mlp.COMPACT_POLICY != "auto" or torch.cuda.is_current_stream_capturing():
            return
        off = self.__dict__.setdefault("_compact_off", {})
        if level in off:
            return
        fills = self.__dict__.setdefault("_compact_fill", {}).setdefault(level, [])
        valid = sum(int(c.gstart[-1]) for c in comps)            # device-to-host: waits for the compaction launch
        fills.append(valid / float(sum(c.cap for c in comps)))
        if len(fills) >= fused_mlp.COMPACT_PROBES:
            off[level] = (sum(fills) / len(fills)) > fused_mlp.COMPACT_MAX_FILL

    def _geometry_prepass(self, xyz, with_ball_csr=False):
        """Everything that depends on coordinates only, on the geometry stream.
        Returns (per-level [new_xyz, [ball idx], event, inverse lists or None], per-FP-level (idx, weight, csr, event),
        event after the inverse lists).  The inverse lists of the ball indices (``with_ball_csr``) are only read by the
        backward pass, so they are built last and the forward never waits for them."""
        gen = self._geometry_prepass_steps(xyz, with_ball_csr)
        try:
            while True:
                next(gen)
        except StopIteration as done:
            return done.value

    def _geometry_prepass_steps(self, xyz, with_ball_csr=False):
        """``_geometry_prepass`` as a generator that yields ONCE, after the level-1 furthest-point sampling has been issued:
        that launch is a 511-round chain on 32 waves -- long, and next to no load on the chip -- while everything after it is
        ~150 us of short bandwidth kernels.  ``prefetch_geometry(..., split=True)`` issues the two parts at different points of
        the step.  No ``with`` block is open across the yield (the caller's current stream and grad mode are its own)."""
        dev = xyz.device
        main, side = torch.cuda.current_stream(dev), _geometry_stream(dev)
        side.wait_stream(main)            # xyz is produced on main; also orders reuse of last step's buffers
        sa_geo, fp_geo = [], [None] * len(self.FP_modules)
        chain = getattr(pointnet2_utils._ext, "furthest_point_sampling_chain", None)     # absent from a plain reference _ext

        def sample(li, sa, cur, tie):
            if chain is not None and cur.shape[1] <= 4096:
                nxt = self.SA_modules[li + 1].npoint if li + 1 < len(self.SA_modules) else 0
                # above FPS_CHAIN_MAX_TRACKED_N points a level scans without the tie bookkeeping (and its child scans too)
                rounds = min(nxt or 0, sa.npoint) if cur.shape[1] <= FPS_CHAIN_MAX_TRACKED_N else 0
                _, new_xyz, tie = chain(cur, sa.npoint, tie_in=tie, track_rounds=rounds)
                return new_xyz, tie
            return sa._sample_centroids(cur), None

        with torch.cuda.stream(side), torch.no_grad():
            first = sample(0, self.SA_modules[0], xyz, None)
        yield
        with torch.cuda.stream(side), torch.no_grad():
            cur, levels = xyz, [xyz]
            tie = None
            for li, sa in enumerate(self.SA_modules):
                new_xyz, tie = first if li == 0 else sample(li, sa, cur, tie)
                ext = pointnet2_utils._ext
                ball_compact = getattr(ext, "ball_compact", None)                    # absent from a plain reference _ext
                compact = (ball_compact is not None and len(sa_geo) in fused_mlp.COMPACT_LEVELS
                           and self._compact_wanted(len(sa_geo)))
                pair = getattr(ext, "ball_query_pair", None) if len(sa.groupers) == 2 else None
                comps = None
                if pair is not None:
                    # both radii from one pass over the cloud, the compact-column counts from the same launch; the column
                    # tables of both scales in 2 launches: 8 -> 3 launches on level 1, 2 -> 1 on the others (bit-identical)
                    ga, gb = sa.groupers
                    ia, ib, glens = pair(new_xyz, cur, (ga.radius, gb.radius), (ga.nsample, gb.nsample), want_glen=compact)
                    idx = [ia, ib]
                    if compact:
                        comps = ext.ball_compact_pair(ia, ib, cur.shape[1], glens)
                else:
                    idx = [pointnet2_utils.ball_query(g.radius, g.nsample, cur, new_xyz) for g in sa.groupers]
                    if compact:
                        comps = [ball_compact(i, cur.shape[1]) for i in idx]
                        comps = comps if all(c is not None for c in comps) else None
                if compact and comps is not None:
                    self._compact_probe(len(sa_geo), comps)
                ev = torch.cuda.Event()
                ev.record(side)
                sa_geo.append([new_xyz, idx, ev, None, comps])
                levels.append(new_xyz)
                cur = new_xyz
            order = list(range(len(self.FP_modules) - 1, -1, -1))
            multi = getattr(pointnet2_utils._ext, "three_nn_weights_multi", None)     # absent from a plain reference _ext
            if multi is not None and len(order) <= 8 and all(levels[l + 1].shape[1] >= 3 for l in order):
                # the neighbour searches of all propagation levels in one launch (coarse level first, as the forward asks)
                res = multi([(levels[l], levels[l + 1]) for l in order])
                ev = torch.cuda.Event()
                ev.record(side)
                for l, (idx, weight) in zip(order, res):
                    fp_geo[l] = [idx, weight, None, ev]
            else:
                for lvl in order:
                    idx, weight = PointnetFPModule.interpolation_weights(levels[lvl], levels[lvl + 1])
                    ev = torch.cuda.Event()
                    ev.record(side)
                    fp_geo[lvl] = [idx, weight, None, ev]
            csr_ev = None
            csr_multi = getattr(pointnet2_utils._ext, "csr_multi", None)     # absent from a plain reference _ext
            if with_ball_csr and csr_multi is not None:
                # inverse lists of every index tensor the backward pass scatters through -- ball queries of the levels
                # that have input features, three_nn taps of every FP level -- in ONE launch, at the end of the
                # pre-pass: only the backward pass reads them
                problems, where = [], []
                if fused_mlp.USE_CSR_SCATTER:
                    for lvl in range(1, len(sa_geo)):       # level 0 has no input features, hence no scatter
                        if sa_geo[lvl][4] is not None:      # compact-column level: lists of its columns instead (below)
                            continue
                        for si, i in enumerate(sa_geo[lvl][1]):
                            problems.append((i, levels[lvl].shape[1]))
                            where.append(("sa", lvl, si))
                for lvl in range(len(self.FP_modules)):
                    problems.append((fp_geo[lvl][0], levels[lvl + 1].shape[1]))
                    where.append(("fp", lvl, 0))
                lists = csr_multi(problems)
                for (kind, lvl, si), res in zip(where, lists):
                    if kind == "fp":
                        fp_geo[lvl][2] = res
                    else:
                        if sa_geo[lvl][3] is None:
                            sa_geo[lvl][3] = [None] * len(sa_geo[lvl][1])
                        sa_geo[lvl][3][si] = res
                for lvl in range(1, len(sa_geo)):
                    if sa_geo[lvl][3] is not None and any(c is None for c in sa_geo[lvl][3]):
                        sa_geo[lvl][3] = None
                # inverse lists of the compact columns of the levels that have input features (their backward scatters
                # the layer-0 gradient through them)
                lists_of = getattr(pointnet2_utils._ext, "ball_compact_lists", None)
                todo = [c for lvl in range(1, len(sa_geo)) for c in (sa_geo[lvl][4] or ())]
                if todo and lists_of is not None and fused_mlp.USE_CSR_SCATTER:
                    lists_of(todo)
                csr_ev = torch.cuda.Event()
                csr_ev.record(side)
            fp_geo = [tuple(g) for g in fp_geo]
        return sa_geo, fp_geo, csr_ev

    def prefetch_geometry(self, pointcloud, slot=None, split=False):
        """Start the coordinate-only work of ``pointcloud`` NOW on the geometry stream -- concurrently with whatever
        the current stream does next, typically the training step of the previous batch -- and keep the results in
        ``slot`` for ``forward(pointcloud, geometry=slot)``.  This is next-batch preprocessing in the sense of a data
        loader: FPS is a 500-round latency chain per level that nothing inside a step can overlap.
        The slot's buffers are allocated on first use and overwritten afterwards (call this once outside a graph
        capture before capturing steps that use the slot).  Returns the slot.

        ``split=True``: only the level-1 furthest-point sampling is issued now; the call returns ``finish``, and
        ``finish()`` -- called later in the step, from the stream whose work the rest should follow -- issues everything
        else (the geometry stream first waits for the calling stream) and returns the slot."""
        gen = self._prefetch_steps(pointcloud, slot)
        if not split:
            try:
                while True:
                    next(gen)
            except StopIteration as done:
                return done.value
        next(gen)

        def finish():
            try:
                next(gen)
            except StopIteration as done:
                return done.value
            raise RuntimeError("prefetch_geometry: the second part did not complete")
        return finish

    def _prefetch_steps(self, pointcloud, slot):
        slot = slot if slot is not None else GeometrySlot()
        xyz, _ = self._break_up_pc(pointcloud)
        if not self._can_prepass(xyz):
            raise RuntimeError("prefetch_geometry needs a CUDA point cloud and plain MSG set-abstraction levels")
        # The ops write straight into the slot's persistent buffers: an _ext.OutputPlan recorded on one pass (plain
        # allocations, copied into the slot as before) hands the k-th output allocation of every later pass its slot view
        # (round 5: the two pack launches per step -- 43 us on the geometry stream -- are gone).  A plain reference _ext has no
        # plan: its results are copied.
        ext = pointnet2_utils._ext
        plan_cls, placing = getattr(ext, "OutputPlan", None), getattr(ext, "placing", None)
        grad_mode = torch.is_grad_enabled()
        stale = slot.sa is None or slot.shape != tuple(xyz.shape) or slot.grad_mode != grad_mode
        plan = None if stale else slot.plan
        recording = plan_cls is not None and plan is None and not torch.cuda.is_current_stream_capturing()
        if recording:
            plan = plan_cls()
        steps = self._geometry_prepass_steps(xyz, with_ball_csr=grad_mode)
        # (the plan is active only while this function runs: whatever the caller issues between the two parts allocates plainly)
        if plan is not None:
            with placing(plan):
                next(steps)
        else:
            next(steps)
        yield
        side = _geometry_stream(xyz.device)
        side.wait_stream(torch.cuda.current_stream(xyz.device))      # split mode: the rest follows the caller's work so far
        try:
            if plan is not None:
                with placing(plan, resume=True):
                    next(steps)
            else:
                next(steps)
            raise RuntimeError("_geometry_prepass_steps yielded twice")
        except StopIteration as done:
            sa_geo, fp_geo, _ = done.value
        with torch.cuda.stream(side), torch.no_grad():
            fresh = GeometrySlot()
            fresh.sa = [(new_xyz, list(idx), csrs, comps) for new_xyz, idx, _, csrs, comps in sa_geo]
            fresh.fp = [(idx, weight, csr) for idx, weight, csr, _ in fp_geo]
            src = fresh.tensors()
            if stale or len(src) != len(slot.tensors()):
                # persistent storage: one flat buffer per dtype, the slot's tensors are views into them
                # every view starts on a 16-byte boundary: kernels read / write some of these tensors with 16-byte accesses
                # (compact_scan_pair_kernel: int4 on the per-row counts and starts; a (b * g + 1)-element table in front of
                # them would otherwise push its successors off alignment -- ADVICE r5)
                def _padded(t):
                    q = max(16 // t.element_size(), 1)
                    return (t.numel() + q - 1) // q * q
                slot.flat = {dt: torch.empty(sum(_padded(t) for t in src if t.dtype == dt), dtype=dt, device=xyz.device)
                             for dt in {t.dtype for t in src}}
                slot.packed = all(_padded(t) == t.numel() for t in src)      # back-to-back: one pack launch can fill it
                off = {dt: 0 for dt in slot.flat}
                views = []
                for t in src:
                    views.append(slot.flat[t.dtype][off[t.dtype]:off[t.dtype] + t.numel()].view(t.shape))
                    off[t.dtype] += _padded(t)
                it = iter(views)
                slot.sa = [(next(it), [next(it) for _ in idxs],
                            [(next(it), next(it)) for _ in csrs] if csrs is not None else None,
                            [c.with_tensors([next(it) for _ in c.tensors()]) for c in comps] if comps is not None else None)
                           for _, idxs, csrs, comps in fresh.sa]
                slot.fp = [(next(it), next(it), (next(it), next(it)) if csr is not None else None)
                           for _, _, csr in fresh.fp]
                slot.shape, slot.grad_mode = tuple(xyz.shape), grad_mode
                slot.plan = None
                if not recording:
                    plan = None          # a plan of another layout placed nothing we can trust: copy, record next time
            views = slot.tensors()
            todo = [(t, v) for t, v in zip(src, views) if t.data_ptr() != v.data_ptr()]
            if todo:
                by_dt = {}
                for t, v in todo:
                    by_dt.setdefault(t.dtype, []).append((t, v))
                for dt, pairs in by_dt.items():
                    whole = len(pairs) == sum(1 for t in src if t.dtype == dt)
                    if (whole and slot.flat[dt].element_size() == 4 and all(t.is_contiguous() for t, _ in pairs)):
                        srcs = [t for t, _ in pairs]
                        if not slot.packed:
                            # the views are 16-byte aligned: a few words of padding follow the odd-sized tensors -- filled
                            # from a zero tensor so that the whole flat buffer is still one pack launch
                            pad = torch.zeros(4, dtype=dt, device=xyz.device)
                            srcs = [piece for t in srcs
                                    for piece in ((t,) if t.numel() % 4 == 0 else (t, pad[:4 - t.numel() % 4]))]
                        _native.pack_words(srcs, slot.flat[dt], side.cuda_stream)   # one launch per <= 64 tensors
                    else:
                        for t, v in pairs:
                            v.copy_(t)
            if recording:
                # from the next pass on, the request that produced src[j] is handed views[j]
                plan.bind({t.data_ptr(): v for t, v in zip(src, views)})
                slot.plan = plan
            if not torch.cuda.is_current_stream_capturing():
                slot.event = torch.cuda.Event()
                slot.event.record(side)
        return slot

    def join_geometry(self, device=None):
        """Make the current stream wait for the geometry stream (end of a pipelined step; required before the
        end of a graph capture that contains a prefetch)."""
        dev = device if device is not None else next(self.parameters()).device
        torch.cuda.current_stream(dev).wait_stream(_geometry_stream(dev))

    def _can_prepass(self, xyz):
        if not (USE_GEOMETRY_STREAM and xyz.is_cuda and not xyz.requires_grad):
            return False
        return all(isinstance(sa, PointnetSAModuleMSG) and sa.npoint is not None
                   and all(type(g) is pointnet2_utils.QueryAndGroup and not g.sample_uniformly for g in sa.groupers)
                   for sa in self.SA_modules)

    def forward(self, pointcloud, geometry=None):
        """(B, N, 3[+C]) -> (B, 128, N).  ``geometry``: a GeometrySlot filled by ``prefetch_geometry`` for THIS
        point cloud (extension of the reference signature)."""
        if (geometry is None and graphed.ENABLED and pointcloud.is_cuda and pointcloud.dim() == 3
                and pointcloud.size(-1) == 3 and pointcloud.dtype == torch.float32 and _native.TIMING is None and _native.MARKERS is None):
            # an eager caller (the reference's loop, utils/solver.py:88-99): forward and backward of a shape seen before are
            # one HIP-graph launch each (graphed.AutoGraph: same kernels, same order, bit-identical results)
            ref = weakref.ref(self)       # (the AutoGraph is the value of a WeakKeyDictionary keyed by this module)

            def state():
                me = ref()
                off = me.__dict__.get("_compact_off", {}) if me is not None else {}
                # + the levels whose compact columns the data switched off (_compact_probe): another kernel sequence
                return _switch_state() + (tuple(sorted(l for l, v in off.items() if v)),)
            return graphed.for_module(self, PointNet2MSG._plain_forward, state)(pointcloud)
        return self._plain_forward(pointcloud, geometry)

    def _plain_forward(self, pointcloud, geometry=None):
        with defer_bn_counters():
            return self._forward(pointcloud, geometry)

    def _forward(self, pointcloud, geometry=None):
        xyz, features = self._break_up_pc(pointcloud)
        sa_geo = fp_geo = csr_ev = None
        if geometry is not None:
            if geometry.sa is None or geometry.shape != tuple(xyz.shape):
                raise RuntimeError("forward(geometry=...): the slot was not prefetched for a cloud of this shape")
            main = torch.cuda.current_stream(xyz.device)
            if geometry.event is not None and not torch.cuda.is_current_stream_capturing():
                main.wait_event(geometry.event)   # inside a capture the prefetching graph has completed already
            sa_geo = [(nx, idx, None, csrs, comps) for nx, idx, csrs, comps in geometry.sa]
            fp_geo = [(i, w, csr, None) for i, w, csr in geometry.fp]
        elif self._can_prepass(xyz):
            sa_geo, fp_geo, csr_ev = self._geometry_prepass(xyz, with_ball_csr=torch.is_grad_enabled())
            main = torch.cuda.current_stream(xyz.device)
        l_xyz, l_features = [xyz], [features]
        _native.mark("fwd start")
        for i, sa in enumerate(self.SA_modules):
            if sa_geo is None:
                nxt_xyz, nxt_feat = sa(l_xyz[-1], l_features[-1])
            else:
                new_xyz, idx, ev, csrs, comps = sa_geo[i]
                if ev is not None:
                    main.wait_event(ev)
                nxt_xyz, nxt_feat = sa(l_xyz[-1], l_features[-1], geometry=(new_xyz, idx, csrs, comps))
            l_xyz.append(nxt_xyz)
            l_features.append(nxt_feat)
            _native.mark(f"fwd SA{i + 1} done")
        if geometry is not None and geometry.after_sa is not None:
            hook, geometry.after_sa = geometry.after_sa, None
            hook()
        for lvl in range(len(self.FP_modules) - 1, -1, -1):  # coarse -> fine  [ref :322-325]
            interp = None
            if fp_geo is not None:
                idx, weight, csr, ev = fp_geo[lvl]
                if ev is not None:
                    main.wait_event(ev)
                interp = (idx, weight, csr)
            # levels 3..1 hand their RAW last output + BatchNorm constants to the next level (fused_mlp.LazyAct): its
            # loaders apply the activation, the tensor of activated features is never written  [ref :322-325]
            lazy = lvl > 0 and type(self.FP_modules[lvl - 1]) is PointnetFPModule
            l_features[lvl] = self.FP_modules[lvl](l_xyz[lvl], l_xyz[lvl + 1], l_features[lvl],
                                                   l_features[lvl + 1], interp=interp, lazy_out=lazy)
            _native.mark(f"fwd FP{lvl + 1} done")
        if csr_ev is not None:
            main.wait_event(csr_ev)       # the backward pass reads the inverse lists built at the end of the pre-pass
        return l_features[0]

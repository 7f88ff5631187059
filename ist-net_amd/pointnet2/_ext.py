"""Drop-in for the reference's pybind module ``pointnet2._ext``.

Same nine free functions, argument order and error behaviour as
model/pointnet2/_ext_src/src/bindings.cpp:11-24 and the four host files
(ball_query.cpp, group_points.cpp, interpolate.cpp, sampling.cpp); the work runs in the
hand-written gfx950 kernels of ``csrc/pn2_index_ops.hip`` through the C ABI of
``include/istnet_pn2.h``.  Outputs are allocated here with ``torch.empty`` (the kernels write
every element; the reference allocates with ``torch::zeros``), launches go to the caller's
current stream on the tensors' device, nothing synchronises.

CPU tensors raise ``RuntimeError("CPU not supported")`` exactly like the reference
(e.g. ball_query.cpp:32-34); there is no fallback of any kind.
"""
import torch

from .. import _native


class OutputPlan:
    """Where the outputs of a fixed SEQUENCE of op calls should live.  ``PointNet2MSG.prefetch_geometry`` refills a
    GeometrySlot (persistent buffers two captured steps exchange) every step; the ops used to write fresh tensors that one or
    two pack launches then copied into the slot.  With a plan active (``placing(plan)``) the k-th output allocation of the
    sequence is handed the slot view recorded for it, so the kernels write the slot directly.  First pass: ``record`` mode,
    plain allocations are remembered in call order; the caller then maps them to its views (``bind``).  A request whose shape or
    dtype differs from the recorded one falls back to a plain allocation (the caller copies whatever is not in place)."""

    def __init__(self):
        self.recorded = []      # record mode: tensors handed out, in call order
        self.views = None       # replay mode: view (or None) per request
        self.k = 0

    def bind(self, placed):
        """``placed``: {data_ptr of a recorded tensor: persistent view it should be written to from now on}."""
        self.views = [placed.get(t.data_ptr()) for t in self.recorded]
        self.views = [v if (v is not None and v.shape == t.shape and v.dtype == t.dtype) else None
                      for v, t in zip(self.views, self.recorded)]
        self.recorded = None

    def take(self, shape, dtype, device):
        if self.views is None:
            t = torch.empty(shape, dtype=dtype, device=device)
            self.recorded.append(t)
            return t
        v = self.views[self.k] if self.k < len(self.views) else None
        self.k += 1
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        if v is not None and tuple(v.shape) == shape and v.dtype == dtype and v.device == torch.device(device):
            return v
        return torch.empty(shape, dtype=dtype, device=device)


_PLAN = None


class placing:
    def __init__(self, plan, resume=False):
        self.plan, self.resume = plan, resume      # resume: continue the sequence where an earlier block left it

    def __enter__(self):
        global _PLAN
        self.prev, _PLAN = _PLAN, self.plan
        if self.plan is not None and not self.resume:
            self.plan.k = 0
        return self.plan

    def __exit__(self, *exc):
        global _PLAN
        _PLAN = self.prev
        return False


def _new(shape, dtype=None, device=None):
    if _PLAN is None:
        return torch.empty(shape, dtype=dtype, device=device)
    return _PLAN.take(shape, dtype, device)



def set_distance_convention(convention):
    """Arithmetic of the index-deciding squared distances in furthest_point_sampling / ball_query / three_nn:
    0 = the reference's source expression ``((dx*dx + dy*dy) + dz*dz)`` with every operation rounded (default);
    1 = ``fma(dz,dz, fma(dx,dx, dy*dy))`` -- what a contracting compiler makes of that expression, i.e. what the reference
        built with its own setup.py (nvcc -O3, default -fmad=true) most likely computes: SELECT THIS to reproduce a CUDA
        build's indices bit for bit (profiles/r05_fma_contraction_llvm.txt);
    2 = ``fma(dz,dz, fma(dy,dy, dx*dx))``, the other possible fusion.
    Process-wide; returns the previous value.  The environment variable ISTNET_DISTANCE_CONVENTION sets it at load time.
    Measured exposure: 0 of 8.6 M index entries on scan-like clouds, 0.008 % on uniform cubes (profiles/r02_fma_convention_flips.txt)."""
    global _CONVENTION
    convention = int(convention)
    if _native.lib().istnet_pn2_set_tuning(1, convention) != 0:
        raise ValueError(f"distance convention must be 0, 1 or 2, got {convention}")
    prev, _CONVENTION = _CONVENTION, convention
    return prev


_CONVENTION = 0


def _req(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _contig(t, name):
    _req(t.is_contiguous(), f"{name} must be a contiguous tensor")      # utils.h:15-18


def _is_float(t, name):
    _req(t.dtype == torch.float32, f"{name} must be a float tensor")    # utils.h:26-30


def _is_int(t, name):
    _req(t.dtype == torch.int32, f"{name} must be an int tensor")       # utils.h:20-24


def _device_of(lead, lead_name, *others):
    """Reference rule: if the leading tensor is CUDA the others must be too; CPU is unsupported."""
    if not lead.is_cuda:
        raise RuntimeError("CPU not supported")
    for t, name in others:
        _req(t.is_cuda, f"{name} must be a CUDA tensor")                # utils.h:10-13
        _req(t.device == lead.device, f"{name} must be on the same device as {lead_name}")
    return lead.device


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _ptr(t):
    return t.data_ptr()


def gather_points(points, idx):
    """(B,C,N) f32, (B,npoint) i32 -> (B,C,npoint).  sampling.cpp:20-43"""
    _contig(points, "points"); _contig(idx, "idx"); _is_float(points, "points"); _is_int(idx, "idx")
    dev = _device_of(points, "points", (idx, "idx"))
    b, c, n = points.shape
    m = idx.shape[1]
    out = _new((b, c, m), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_pn2_gather_points(
            b, c, n, m, _ptr(points), _ptr(idx), _ptr(out), _stream(dev)), "gather_points")
    return out


def gather_points_grad(grad_out, idx, n):
    """(B,C,npoint) f32, (B,npoint) i32, n -> (B,C,n).  sampling.cpp:45-68"""
    _contig(grad_out, "grad_out"); _contig(idx, "idx"); _is_float(grad_out, "grad_out"); _is_int(idx, "idx")
    dev = _device_of(grad_out, "grad_out", (idx, "idx"))
    b, c, m = grad_out.shape
    out = _new((b, c, int(n)), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_pn2_gather_points_grad(
            b, c, int(n), m, _ptr(grad_out), _ptr(idx), _ptr(out), _stream(dev)), "gather_points_grad")
    return out


def furthest_point_sampling(points, nsamples):
    """(B,N,3) f32 -> (B,nsamples) i32, first index 0.  sampling.cpp:70-91"""
    _contig(points, "points"); _is_float(points, "points")
    dev = _device_of(points, "points")
    b, n = points.shape[0], points.shape[1]
    nsamples = int(nsamples)
    out = _new((b, nsamples), dtype=torch.int32, device=dev)
    # scratch is only needed by the large-cloud kernel (n > 4096); see include/istnet_pn2.h
    tmp = _new((b, n), dtype=torch.float32, device=dev) if n > 4096 else None
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_pn2_furthest_point_sampling(
            b, n, nsamples, _ptr(points), _ptr(tmp) if tmp is not None else None, _ptr(out),
            _stream(dev)), "furthest_point_sampling")
    return out


def furthest_point_sampling_gather(points, nsamples):
    """(B,N,3) f32 -> idx (B,nsamples) i32 and the picked coordinates (B,nsamples,3) in one launch
    (extension: the reference gathers them with gather_points on the transposed cloud).  N <= 4096."""
    _contig(points, "points"); _is_float(points, "points")
    dev = _device_of(points, "points")
    b, n = points.shape[0], points.shape[1]
    nsamples = int(nsamples)
    out = _new((b, nsamples), dtype=torch.int32, device=dev)
    picked = _new((b, nsamples, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_pn2_fps_gather(b, n, nsamples, _ptr(points), _ptr(out), _ptr(picked),
                                                          _stream(dev)), "fps_gather")
    return out, picked


def furthest_point_sampling_chain(points, nsamples, tie_in=None, track_rounds=None):
    """Sampling for a stack of set-abstraction levels: (idx (B,nsamples) i32, picked (B,nsamples,3), tie (B,) i32).
    ``tie`` = first round of this run whose maximum was not unique (INT_MAX if none, ``track_rounds`` if only the
    rounds below it were examined).  ``tie_in`` = the ``tie`` of the run whose picks, in pick order, ARE ``points``:
    clouds with ``tie_in >= nsamples`` get the prefix 0..nsamples-1 without a scan (include/istnet_pn2.h).  Results
    equal furthest_point_sampling_gather bit for bit.  N <= 4096."""
    _contig(points, "points"); _is_float(points, "points")
    dev = _device_of(points, "points")
    b, n = points.shape[0], points.shape[1]
    nsamples = int(nsamples)
    if tie_in is not None:
        _contig(tie_in, "tie_in"); _is_int(tie_in, "tie_in")
        _req(tie_in.is_cuda and tie_in.device == dev and tie_in.numel() == b, "tie_in must be (B,) int32 on the device of points")
    out = _new((b, nsamples), dtype=torch.int32, device=dev)
    picked = _new((b, nsamples, 3), dtype=torch.float32, device=dev)
    tie = _new((b,), dtype=torch.int32, device=dev)
    track = nsamples if track_rounds is None else int(track_rounds)
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_pn2_fps_gather_chain(
            b, n, nsamples, _ptr(points), _ptr(out), _ptr(picked), _ptr(tie_in) if tie_in is not None else None,
            _ptr(tie), track, _stream(dev)), "fps_gather_chain")
    return out, picked, tie


def three_nn(unknowns, knows):
    """(B,n,3), (B,m,3) f32 -> [dist2 (B,n,3) f32, idx (B,n,3) i32].  interpolate.cpp:19-45"""
    _contig(unknowns, "unknowns"); _contig(knows, "knows")
    _is_float(unknowns, "unknowns"); _is_float(knows, "knows")
    dev = _device_of(unknowns, "unknowns", (knows, "knows"))
    b, n = unknowns.shape[0], unknowns.shape[1]
    m = knows.shape[1]
    idx = _new((b, n, 3), dtype=torch.int32, device=dev)
    dist2 = _new((b, n, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_pn2_three_nn(
            b, n, m, _ptr(unknowns), _ptr(knows), _ptr(dist2), _ptr(idx), _stream(dev)), "three_nn")
    return [dist2, idx]


def three_nn_weights(unknowns, knows):
    """(B,n,3), (B,m,3) f32 -> (idx (B,n,3) i32, weight (B,n,3) f32): three_nn and the inverse-distance weights of
    PointnetFPModule (pointnet2_modules.py:185-188) in one launch (extension of the reference's nine functions)."""
    _contig(unknowns, "unknowns"); _contig(knows, "knows")
    _is_float(unknowns, "unknowns"); _is_float(knows, "knows")
    dev = _device_of(unknowns, "unknowns", (knows, "knows"))
    b, n = unknowns.shape[0], unknowns.shape[1]
    m = knows.shape[1]
    idx = _new((b, n, 3), dtype=torch.int32, device=dev)
    weight = _new((b, n, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_pn2_three_nn_weights(
            b, n, m, _ptr(unknowns), _ptr(knows), _ptr(idx), _ptr(weight), _stream(dev)), "three_nn_weights")
    return idx, weight


def three_nn_weights_multi(pairs):
    """``three_nn_weights`` for several (unknowns, knows) pairs over the same B clouds in one launch (the propagation levels
    of an encoder pass); a list of (idx, weight), each bit-identical to the stand-alone call.  At most 8 pairs."""
    import ctypes
    _req(0 < len(pairs) <= 8, "1 to 8 problems")
    dev = None
    for u, k in pairs:
        _contig(u, "unknowns"); _contig(k, "knows"); _is_float(u, "unknowns"); _is_float(k, "knows")
        d = _device_of(u, "unknowns", (k, "knows"))
        _req(dev is None or d == dev, "all problems on one device")
        _req(u.shape[0] == pairs[0][0].shape[0] == k.shape[0], "all problems over the same clouds")
        dev = d
    b = pairs[0][0].shape[0]
    outs = [(_new((b, u.shape[1], 3), dtype=torch.int32, device=dev), _new((b, u.shape[1], 3), dtype=torch.float32, device=dev))
            for u, _ in pairs]
    k = len(pairs)
    ints = lambda vals: (ctypes.c_int * k)(*vals)
    ptrs = lambda ts: (ctypes.c_void_p * k)(*[_ptr(t) for t in ts])
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_pn2_three_nn_weights_multi(
            k, b, ints([u.shape[1] for u, _ in pairs]), ints([kn.shape[1] for _, kn in pairs]), ptrs([u for u, _ in pairs]),
            ptrs([kn for _, kn in pairs]), ptrs([o[0] for o in outs]), ptrs([o[1] for o in outs]), _stream(dev)),
            "three_nn_weights_multi")
    return outs


def three_interpolate(points, idx, weight):
    """(B,C,m) f32, (B,n,3) i32, (B,n,3) f32 -> (B,C,n).  interpolate.cpp:47-74"""
    _contig(points, "points"); _contig(idx, "idx"); _contig(weight, "weight")
    _is_float(points, "points"); _is_int(idx, "idx"); _is_float(weight, "weight")
    dev = _device_of(points, "points", (idx, "idx"), (weight, "weight"))
    b, c, m = points.shape
    n = idx.shape[1]
    out = _new((b, c, n), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_pn2_three_interpolate(
            b, c, m, n, _ptr(points), _ptr(idx), _ptr(weight), _ptr(out), _stream(dev)),
            "three_interpolate")
    return out


def _csr_fits(e, m):
    """(range kernel ok, legacy kernel ok) for e slots per cloud with keys in [0, m): mirrors the launchers' LDS rules
    (csr_range_lds / the one-workgroup-per-cloud histogram) in csrc/pn2_index_ops.hip."""
    qcap = ((e + 3) // 4 + 63) // 64 * 64 + 8
    return (4 * qcap + 4 * 64 + 4) * 4 <= 65536 and e < (1 << 24), 3 * m + 257 <= 16384


def csr_multi(problems):
    """Inverse lists of several index tensors over the same batch in ONE launch (istnet_pn2_csr_build_multi).

    problems: list of (idx, m) with idx an int32 CUDA tensor (B, ...) of keys in [0, m); every trailing dim is
    flattened into the slot index e.  Returns a list of (offsets (B, m+1), entries (B, E)) int32 -- entries of a list in
    ascending slot order -- or None for a problem too large for either build kernel."""
    import ctypes
    out, batch = [None] * len(problems), []
    lib = _native.lib()
    dev = None
    for i, (idx, m) in enumerate(problems):
        _contig(idx, "idx"); _is_int(idx, "idx")
        dev = _device_of(idx, "idx")
        b, m = idx.shape[0], int(m)
        e = idx.numel() // max(b, 1)
        fits_range, fits_legacy = _csr_fits(e, m)
        if not (fits_range or fits_legacy):
            continue
        offsets = _new((b, m + 1), dtype=torch.int32, device=dev)
        entries = _new((b, e), dtype=torch.int32, device=dev)
        out[i] = (offsets, entries)
        if fits_range:
            batch.append((b, e, m, idx, offsets, entries))
        else:
            with torch.cuda.device(dev):
                _native.check(lib.istnet_pn2_csr_build(b, e, m, _ptr(idx), _ptr(offsets), _ptr(entries), _stream(dev)),
                              "csr_build")
    for b in sorted({t[0] for t in batch}):
        group = [t for t in batch if t[0] == b]
        for i in range(0, len(group), 12):
            chunk = group[i:i + 12]
            n = len(chunk)
            arr = lambda k: (ctypes.c_void_p * n)(*[_ptr(t[k]) for t in chunk])
            with torch.cuda.device(dev):
                _native.check(lib.istnet_pn2_csr_build_multi(
                    n, b, (ctypes.c_int * n)(*[t[1] for t in chunk]), (ctypes.c_int * n)(*[t[2] for t in chunk]),
                    arr(3), arr(4), arr(5), _stream(dev)), "csr_build_multi")
    return out


def interp_csr(idx, m):
    """Per-cloud inverse lists of a three_nn index tensor (B,n,3) over m sources: (offsets (B,m+1), entries (B,3n))
    int32 -- tap e = 3*j + t, ascending inside each list -- or None when the tensor is too large for the build kernels.
    Depends on idx only, so a caller that knows idx early (the encoder's geometry pre-pass) can build it off the
    critical path and hand it to three_interpolate_grad."""
    return csr_multi([(idx, m)])[0]


def ball_csr(idx, n):
    """Per-cloud inverse lists of a ball-query index tensor (B,npoint,nsample) over the n source points:
    (offsets (B,n+1), entries (B,npoint*nsample)) int32, entries of a list in ascending slot order; None when the
    tensor is too large for the build kernels.  Depends on coordinates only (build it in the geometry pre-pass);
    consumed by the atomic-free layer-0 gradient scatter of the fused set-abstraction backward."""
    return csr_multi([(idx, n)])[0]


class BallCompact:
    """Compact-column tables of one ball-query index tensor (csrc/sa_compact.hip): per group its distinct neighbours
    plus one weighted representative of the padded repeats, all groups of all clouds on one point axis of static
    capacity ``cap`` = B * npoint * nsample; the valid column count is ``gstart[-1]`` on the device.
    ``cstart`` (B + 1) = first column of every cloud; ``csr`` = (offsets (B, n + 1), entries (cap)) inverse lists of the
    columns over the source points (``ball_compact_lists``), needed by the backward of a scale that has input features."""
    __slots__ = ("glen", "gstart", "cidx", "meta", "colw", "cstart", "csr", "b", "g", "s", "n", "cap")

    def __init__(self, glen, gstart, cidx, meta, colw, b, g, s, n, cstart=None, csr=None):
        self.glen, self.gstart, self.cidx, self.meta, self.colw = glen, gstart, cidx, meta, colw
        self.cstart, self.csr = cstart, csr
        self.b, self.g, self.s, self.n, self.cap = b, g, s, n, b * g * s

    @property
    def ncols_ptr(self):
        return self.gstart.data_ptr() + 4 * self.b * self.g

    def tensors(self):
        out = [self.glen, self.gstart, self.cidx, self.meta, self.colw]
        if self.csr is not None:
            out += [self.cstart, *self.csr]
        return out

    def with_tensors(self, tensors):
        tensors = list(tensors)
        cm = BallCompact(*tensors[:5], self.b, self.g, self.s, self.n)
        if self.csr is not None:
            cm.cstart, cm.csr = tensors[5], (tensors[6], tensors[7])
        return cm


def ball_compact_lists(compacts):
    """Inverse lists (source point -> its compact columns, ascending) for several BallCompact tables in one launch
    (istnet_pn2_csr_build_segmented); sets ``cstart`` / ``csr`` on each and returns True, or False when a table is too
    large for the build kernel (nothing is set then)."""
    import ctypes
    compacts = [c for c in compacts if c is not None]
    if not compacts:
        return False
    lib = _native.lib()
    b = compacts[0].b
    if any(c.b != b for c in compacts) or len(compacts) > 12:
        return False
    if not all(_csr_fits(c.g * c.s, c.n)[0] for c in compacts):
        return False
    dev = compacts[0].gstart.device
    work = []
    for c in compacts:
        cstart = _new((b + 1,), dtype=torch.int32, device=dev)                  # first column of every cloud
        cstart.copy_(c.gstart[::c.g])
        off = _new((b, c.n + 1), dtype=torch.int32, device=dev)
        ent = _new((c.cap,), dtype=torch.int32, device=dev)
        work.append((c, cstart, off, ent))
    n = len(work)
    arr = lambda ts: (ctypes.c_void_p * n)(*[_ptr(t) for t in ts])
    with torch.cuda.device(dev):
        _native.check(lib.istnet_pn2_csr_build_segmented(
            n, b, (ctypes.c_int * n)(*[c.g * c.s for c, _, _, _ in work]), (ctypes.c_int * n)(*[c.n for c, _, _, _ in work]),
            arr([c.cidx for c, _, _, _ in work]), arr([w[2] for w in work]), arr([w[3] for w in work]),
            arr([w[1] for w in work]), (ctypes.c_int * n)(*[c.n for c, _, _, _ in work]), _stream(dev)),
            "csr_build_segmented")
    for c, cstart, off, ent in work:
        c.cstart, c.csr = cstart, (off, ent)
    return True


def ball_compact(idx, n):
    """Column tables for the compact evaluation of a set-abstraction scale from its ball-query indices
    (B, npoint, nsample) over n source points per cloud; None when the shape is outside what the kernels take
    (capacity must be a multiple of 256, nsample <= 64).  Depends on coordinates only."""
    _contig(idx, "idx"); _is_int(idx, "idx")
    dev = _device_of(idx, "idx")
    b, g, s = idx.shape
    cap = b * g * s
    if cap % 256 or s > 64 or b * g > 1024 * 64 or cap >= 2 ** 31:
        return None
    glen = _new(b * g, dtype=torch.int32, device=dev)
    gstart = _new(b * g + 1, dtype=torch.int32, device=dev)
    cidx = _new(cap, dtype=torch.int32, device=dev)
    meta = _new(cap, dtype=torch.int32, device=dev)
    colw = _new(cap, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_sa_compact(b, g, s, int(n), _ptr(idx), _ptr(glen), _ptr(gstart), _ptr(cidx),
                                                      _ptr(meta), _ptr(colw), _stream(dev)), "sa_compact")
    return BallCompact(glen, gstart, cidx, meta, colw, b, g, s, int(n))


def three_interpolate_grad(grad_out, idx, weight, m, csr=None):
    """(B,C,n) f32, (B,n,3) i32, (B,n,3) f32, m -> (B,C,m).  interpolate.cpp:75-104
    ``csr``: optional result of interp_csr(idx, m) (extension of the reference signature)."""
    _contig(grad_out, "grad_out"); _contig(idx, "idx"); _contig(weight, "weight")
    _is_float(grad_out, "grad_out"); _is_int(idx, "idx"); _is_float(weight, "weight")
    dev = _device_of(grad_out, "grad_out", (idx, "idx"), (weight, "weight"))
    b, c, n = grad_out.shape
    m = int(m)
    out = _new((b, c, m), dtype=torch.float32, device=dev)
    lib = _native.lib()
    if csr is None:
        csr = interp_csr(idx, m)   # deterministic gather over per-cloud inverse lists, reused by all channels
    with torch.cuda.device(dev):
        if csr is not None:
            offsets, entries = csr
            _native.check(lib.istnet_pn2_three_interpolate_grad_csr(
                b, c, n, m, _ptr(grad_out), _ptr(weight), _ptr(offsets), _ptr(entries), _ptr(out), _stream(dev)),
                "three_interpolate_grad_csr")
        else:
            _native.check(lib.istnet_pn2_three_interpolate_grad(
                b, c, n, m, _ptr(grad_out), _ptr(idx), _ptr(weight), _ptr(out), _stream(dev)),
                "three_interpolate_grad")
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """(B,npoint,3), (B,N,3) f32, radius, nsample -> (B,npoint,nsample) i32.  ball_query.cpp:13-37"""
    _contig(new_xyz, "new_xyz"); _contig(xyz, "xyz"); _is_float(new_xyz, "new_xyz"); _is_float(xyz, "xyz")
    dev = _device_of(new_xyz, "new_xyz", (xyz, "xyz"))
    b, m = new_xyz.shape[0], new_xyz.shape[1]
    n = xyz.shape[1]
    nsample = int(nsample)
    idx = _new((b, m, nsample), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_pn2_query_ball_point(
            b, n, m, float(radius), nsample, _ptr(new_xyz), _ptr(xyz), _ptr(idx), _stream(dev)),
            "ball_query")
    return idx


def ball_query_pair(new_xyz, xyz, radii, nsamples, want_glen=False):
    """The two ball queries of an MSG level (same centroids, two radii) in one pass over the cloud: (idx_a, idx_b, glens) with
    idx_* bit-identical to ``ball_query(new_xyz, xyz, radius, nsample)`` per radius.  ``glens`` = (glen_a, glen_b), the
    compact-column counts ``ball_compact_pair`` starts from, or None.  Extension of the reference's op set
    (ball_query.cpp:13-37 once per radius)."""
    _contig(new_xyz, "new_xyz"); _contig(xyz, "xyz"); _is_float(new_xyz, "new_xyz"); _is_float(xyz, "xyz")
    dev = _device_of(new_xyz, "new_xyz", (xyz, "xyz"))
    b, m = new_xyz.shape[0], new_xyz.shape[1]
    n = xyz.shape[1]
    (ra, rb), (sa, sb) = radii, (int(nsamples[0]), int(nsamples[1]))
    _req(sa > 0 and sb > 0, "nsample must be positive")
    idx_a = _new((b, m, sa), dtype=torch.int32, device=dev)
    idx_b = _new((b, m, sb), dtype=torch.int32, device=dev)
    glens = None
    if want_glen:
        glens = (_new(b * m, dtype=torch.int32, device=dev), _new(b * m, dtype=torch.int32, device=dev))
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_pn2_query_ball_point_pair(
            b, n, m, float(ra), sa, float(rb), sb, _ptr(new_xyz), _ptr(xyz), _ptr(idx_a), _ptr(idx_b),
            _ptr(glens[0]) if glens else None, _ptr(glens[1]) if glens else None, _stream(dev)), "ball_query_pair")
    return idx_a, idx_b, glens


def ball_compact_pair(idx_a, idx_b, n, glens=None):
    """``ball_compact`` for the two scales of one level in 2 launches instead of 6 (3 when ``glens`` -- the counts
    ``ball_query_pair`` wrote -- is None); a list of two BallCompact, or None when a shape is outside what the kernels take."""
    for t, name in ((idx_a, "idx_a"), (idx_b, "idx_b")):
        _contig(t, name); _is_int(t, name)
    dev = _device_of(idx_a, "idx_a", (idx_b, "idx_b"))
    b, g, sa = idx_a.shape
    sb = idx_b.shape[2]
    _req(idx_b.shape[0] == b and idx_b.shape[1] == g, "the two index tensors must share (B, npoint)")
    if any((b * g * s) % 256 or s > 64 or b * g * s >= 2 ** 31 for s in (sa, sb)) or b * g > 1024 * 64:
        return None
    out = []
    for s, gl in ((sa, glens[0] if glens else None), (sb, glens[1] if glens else None)):
        cap = b * g * s
        out.append(BallCompact(gl if gl is not None else _new(b * g, dtype=torch.int32, device=dev),
                               _new(b * g + 1, dtype=torch.int32, device=dev),
                               _new(cap, dtype=torch.int32, device=dev), _new(cap, dtype=torch.int32, device=dev),
                               _new(cap, dtype=torch.float32, device=dev), b, g, s, int(n)))
    ca, cb = out
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_sa_compact_pair(
            b, g, int(n), sa, _ptr(idx_a), _ptr(ca.glen), _ptr(ca.gstart), _ptr(ca.cidx), _ptr(ca.meta), _ptr(ca.colw),
            sb, _ptr(idx_b), _ptr(cb.glen), _ptr(cb.gstart), _ptr(cb.cidx), _ptr(cb.meta), _ptr(cb.colw),
            1 if glens else 0, _stream(dev)), "sa_compact_pair")
    return out


def group_points(points, idx):
    """(B,C,N) f32, (B,npoint,nsample) i32 -> (B,C,npoint,nsample).  group_points.cpp:17-40"""
    _contig(points, "points"); _contig(idx, "idx"); _is_float(points, "points"); _is_int(idx, "idx")
    dev = _device_of(points, "points", (idx, "idx"))
    b, c, n = points.shape
    npoints, nsample = idx.shape[1], idx.shape[2]
    out = _new((b, c, npoints, nsample), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().istnet_pn2_group_points(
            b, c, n, npoints, nsample, _ptr(points), _ptr(idx), _ptr(out), _stream(dev)),
            "group_points")
    return out


def group_points_grad(grad_out, idx, n, csr=None):
    """(B,C,npoint,nsample) f32, (B,npoint,nsample) i32, n -> (B,C,n).  group_points.cpp:42-65
    ``csr``: optional result of ball_csr(idx, n) (extension of the reference signature)."""
    _contig(grad_out, "grad_out"); _contig(idx, "idx"); _is_float(grad_out, "grad_out"); _is_int(idx, "idx")
    dev = _device_of(grad_out, "grad_out", (idx, "idx"))
    b, c, npoints, nsample = grad_out.shape
    out = _new((b, c, int(n)), dtype=torch.float32, device=dev)
    lib = _native.lib()
    # default up to 4096 slots per cloud: deterministic gather over per-cloud inverse lists (one extra launch builds
    # them, all channels reuse them): 2.2-2.6x faster than the LDS-atomic kernel there (list build included); at 8192
    # slots the build (34 us) + a 2-channel-per-workgroup gather (44 us) lose to the atomics (62 us), so larger rows
    # take the list route only when the caller hands the lists in (built once, off the critical path)
    if csr is None and b * c > 0 and 0 < npoints * nsample <= 4096:
        csr = ball_csr(idx, int(n))
    with torch.cuda.device(dev):
        if csr is not None and 0 < npoints * nsample <= 16384:
            _native.check(lib.istnet_pn2_group_points_grad_csr(
                b, c, int(n), npoints, nsample, _ptr(grad_out), _ptr(csr[0]), _ptr(csr[1]), _ptr(out), _stream(dev)),
                "group_points_grad_csr")
        else:
            _native.check(lib.istnet_pn2_group_points_grad(
                b, c, int(n), npoints, nsample, _ptr(grad_out), _ptr(idx), _ptr(out), _stream(dev)),
                "group_points_grad")
    return out


import os as _os
if _os.environ.get("ISTNET_DISTANCE_CONVENTION", "0") != "0" and _os.path.exists(_native.LIB_PATH):
    set_distance_convention(int(_os.environ["ISTNET_DISTANCE_CONVENTION"]))

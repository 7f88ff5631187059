"""GPU parity: every C-ABI op (through the pointnet2._ext drop-in) against the CPU oracle.

Index outputs must be bit-exact; float outputs of pure gathers are bit-exact, of
scatter-adds within 1e-4 (fp32 summation order differs, as it does in the reference's atomics).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _cloud(b, n, seed, kind="cube", scale=1.0):
    g = torch.Generator().manual_seed(seed)
    if kind == "cube":
        return torch.rand(b, n, 3, generator=g) * scale
    if kind == "shell":
        d = torch.randn(b, n, 3, generator=g)
        d = d / d.norm(dim=2, keepdim=True)
        return (d * 0.1 + torch.randn(b, n, 3, generator=g) * 0.002).contiguous()
    if kind == "dup":  # many duplicate points -> FPS ties
        base = torch.rand(b, max(n // 8, 1), 3, generator=g)
        sel = torch.randint(0, base.shape[1], (b, n), generator=g)
        return torch.gather(base, 1, sel.unsqueeze(-1).expand(b, n, 3)).contiguous()
    if kind == "grid":  # lattice -> exact distance ties
        pts = torch.randint(0, 4, (b, n, 3), generator=g).float() * 0.25
        return pts.contiguous()
    raise ValueError(kind)


@pytest.mark.parametrize("b,n,m,kind", [
    (4, 1024, 512, "cube"), (2, 512, 256, "shell"), (2, 256, 128, "cube"), (2, 128, 64, "cube"),
    (3, 1024, 512, "dup"), (2, 1024, 300, "grid"), (2, 1000, 333, "cube"), (2, 700, 100, "dup"),
    (1, 100, 50, "grid"), (2, 65, 64, "cube"), (2, 64, 64, "grid"), (1, 3, 3, "cube"), (1, 1, 1, "cube"),
    (2, 2048, 512, "cube"), (1, 4096, 256, "dup"), (1, 3000, 200, "grid"), (1, 5000, 64, "cube"),
    (1, 1024, 1, "cube"), (1, 37, 80, "cube"),
])
def test_fps_bit_exact(ext, oracle, b, n, m, kind):
    xyz = _cloud(b, n, seed=n * 7 + m, kind=kind)
    want = oracle.furthest_point_sampling(xyz, m)
    got = ext.furthest_point_sampling(xyz.to(DEV), m).cpu()
    assert got.dtype == torch.int32 and got.shape == (b, m)
    assert torch.equal(got, want)
    assert (got[:, 0] == 0).all()


@pytest.mark.parametrize("waves", [1, 2, 4, 8])
def test_fps_multiwave_forms_bit_exact(ext, oracle, waves):
    """The multi-wave form of the register FPS kernel (2 / 4 / 8 waves per cloud; the cross-wave combination is a DPP butterfly
    over the waves' records since round 6) forced for clouds the one-wave kernel normally takes: picks bit-exact against the oracle
    on plain, duplicate-heavy and lattice clouds, and the chained form (tie tracking across waves) equal to level-by-level scans."""
    from istnet_amd import _native
    lib = _native.lib()
    assert lib.istnet_pn2_set_tuning(3, 3) != 0                              # validated knob
    if waves == 1:
        assert lib.istnet_pn2_set_tuning(0, 1025) == 0          # one wave up to 1 024 slots (the default until round 6)
    else:
        assert lib.istnet_pn2_set_tuning(0, 1) == 0 and lib.istnet_pn2_set_tuning(3, waves) == 0
    try:
        for (b, n, m, kind) in [(3, 1024, 512, "cube"), (2, 1024, 512, "shell"), (3, 1024, 512, "dup"), (2, 1024, 300, "grid"),
                                (2, 512, 256, "dup"), (2, 1000, 333, "cube"), (1, 700, 100, "grid"), (2, 2048, 512, "dup")]:
            xyz = _cloud(b, n, seed=n * 11 + m + waves, kind=kind)
            want = oracle.furthest_point_sampling(xyz, m)
            got = ext.furthest_point_sampling(xyz.to(DEV), m).cpu()
            assert torch.equal(got, want), (waves, b, n, m, kind)
        for kind in ("shell", "dup", "grid"):
            xyz = _cloud(4, 1024, seed=5 + waves, kind=kind).to(DEV)
            cur, tie, plain = xyz, None, xyz
            for li, m in enumerate((512, 256, 128, 64)):
                nxt = (256, 128, 64, 0)[li]
                _, cur, tie = ext.furthest_point_sampling_chain(cur, m, tie_in=tie, track_rounds=min(nxt, m))
                _, plain = ext.furthest_point_sampling_gather(plain, m)
                assert torch.equal(cur, plain), (waves, kind, m)
    finally:
        assert lib.istnet_pn2_set_tuning(0, 1024) == 0 and lib.istnet_pn2_set_tuning(3, 4) == 0          # the defaults


@pytest.mark.parametrize("b,n,m,radius,nsample,kind", [
    (4, 1024, 512, 0.2, 32, "cube"),      # BASELINE config 1
    (4, 1024, 512, 0.1, 16, "cube"), (2, 512, 256, 0.02, 16, "shell"), (2, 512, 256, 0.04, 32, "shell"),
    (2, 128, 64, 0.16, 32, "shell"), (2, 1024, 512, 0.25, 32, "grid"), (2, 1000, 77, 0.3, 5, "cube"),
    (1, 64, 64, 10.0, 100, "cube"), (1, 10, 3, 0.5, 4, "cube"), (2, 300, 300, 1e-6, 8, "cube"),
    (1, 6000, 64, 0.05, 64, "cube"), (1, 2048, 700, 0.08, 33, "cube"),
])
def test_ball_query_bit_exact(ext, oracle, b, n, m, radius, nsample, kind):
    xyz = _cloud(b, n, seed=n + m, kind=kind)
    fps = oracle.furthest_point_sampling(xyz, m).long()
    new_xyz = torch.gather(xyz, 1, fps.unsqueeze(-1).expand(b, m, 3)).contiguous()
    if kind == "grid":  # off-lattice centroids too, so that some balls are empty
        new_xyz[:, ::3] += 5.0
    want = oracle.ball_query(new_xyz, xyz, radius, nsample)
    got = ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), radius, nsample).cpu()
    assert got.dtype == torch.int32
    assert torch.equal(got, want)


def test_ball_query_edges(ext, oracle):
    # no neighbour -> zeros; exactly one hit -> row of that index; d2 == r2 excluded
    xyz = torch.tensor([[[0., 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3]]])
    new_xyz = torch.tensor([[[10., 10, 10], [1, 0, 0], [0, 0, 0.5]]])
    for r, ns in [(0.5, 4), (1.0, 3), (2.0, 2), (100.0, 6)]:
        want = oracle.ball_query(new_xyz, xyz, r, ns)
        got = ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), r, ns).cpu()
        assert torch.equal(got, want), (r, ns)
    got = ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), 0.5, 4).cpu()
    assert got[0, 0].tolist() == [0, 0, 0, 0]
    assert got[0, 1].tolist() == [1, 1, 1, 1]
    # point at exactly d2 == r2 is excluded (strict <): centroid (1,0,0), point (0,0,0), r = 1
    got = ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), 1.0, 3).cpu()
    assert got[0, 1].tolist() == [1, 1, 1]


@pytest.mark.parametrize("b,n,m,ra,sa,rb,sb,kind", [
    (2, 1024, 512, 0.01, 16, 0.02, 32, "shell"), (2, 512, 256, 0.02, 16, 0.04, 32, "shell"), (3, 300, 77, 0.3, 8, 0.1, 64, "cube"),
    (2, 256, 64, 0.26, 16, 0.5, 32, "grid"), (1, 100, 40, 0.2, 4, 0.2, 4, "dup"), (2, 128, 64, 0.0005, 16, 0.4, 32, "cube"),
])
def test_ball_query_pair_equals_two_queries(ext, oracle, b, n, m, ra, sa, rb, sb, kind):
    """One pass over the cloud for the two radii of an MSG level: both index tensors bit-identical to the oracle
    (ball_query_gpu.cu:14-49) and to the single-radius launches, either radius larger, lists that fill at different rounds,
    empty balls; the column counts it leaves equal those of the stand-alone compaction."""
    xyz = _cloud(b, n, seed=7 * n + m, kind=kind)
    fps = oracle.furthest_point_sampling(xyz, m).long()
    new_xyz = torch.gather(xyz, 1, fps.unsqueeze(-1).expand(b, m, 3)).contiguous()
    if kind == "grid":
        new_xyz[:, ::3] += 5.0          # empty balls: rows of zeros
    ia, ib, glens = ext.ball_query_pair(new_xyz.to(DEV), xyz.to(DEV), (ra, rb), (sa, sb), want_glen=True)
    for got, r, ns, gl in ((ia, ra, sa, glens[0]), (ib, rb, sb, glens[1])):
        assert torch.equal(got.cpu(), oracle.ball_query(new_xyz, xyz, r, ns))
        assert torch.equal(got, ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), r, ns))
        row = got.cpu()
        cnt = 1 + (row[:, :, 1:] != row[:, :, :1]).sum(-1)
        assert torch.equal(gl.cpu().view(b, m), (cnt + (cnt < ns)).int())
    ja, jb, none = ext.ball_query_pair(new_xyz.to(DEV), xyz.to(DEV), (ra, rb), (sa, sb))
    assert none is None and torch.equal(ja, ia) and torch.equal(jb, ib)


@pytest.mark.parametrize("with_glen", [True, False])
def test_ball_compact_pair_equals_two_compactions(ext, with_glen):
    b, n, m = 4, 1024, 512
    xyz = _cloud(b, n, seed=11, kind="shell").to(DEV)
    new_xyz = xyz[:, :m].contiguous()
    ia, ib, glens = ext.ball_query_pair(new_xyz, xyz, (0.01, 0.02), (16, 32), want_glen=with_glen)
    pair = ext.ball_compact_pair(ia, ib, n, glens)
    for cm, idx in zip(pair, (ia, ib)):
        ref = ext.ball_compact(idx, n)
        t = int(ref.gstart[-1])
        assert int(cm.gstart[-1]) == t and 0 < t < cm.cap
        assert torch.equal(cm.glen, ref.glen) and torch.equal(cm.gstart, ref.gstart)
        up = (t + 255) // 256 * 256
        for a, r in ((cm.cidx, ref.cidx), (cm.meta, ref.meta), (cm.colw, ref.colw)):
            assert torch.equal(a[:up], r[:up])


@pytest.mark.parametrize("b,n,m,kind", [
    (2, 128, 64, "cube"), (2, 256, 128, "shell"), (2, 512, 256, "cube"), (4, 1024, 512, "shell"),
    (2, 1024, 512, "grid"), (2, 333, 77, "dup"), (1, 50, 2, "cube"), (1, 50, 1, "cube"),
    (1, 10, 3, "grid"), (1, 300, 5000, "cube"),
])
def test_three_nn_bit_exact(ext, oracle, b, n, m, kind):
    unknown = _cloud(b, n, seed=3 * n + m, kind=kind)
    known = _cloud(b, m, seed=5 * n + m + 1, kind=kind)
    d_want, i_want = oracle.three_nn(unknown, known)
    d_got, i_got = ext.three_nn(unknown.to(DEV), known.to(DEV))
    assert torch.equal(i_got.cpu(), i_want)
    assert torch.equal(d_got.cpu(), d_want)  # same f32 arithmetic, inf where m < 3


def test_three_nn_weights_multi_equals_single_launches(ext, oracle):
    """All propagation levels in one launch: every problem bit-identical to its own three_nn_weights launch (and its indices to
    the oracle, interpolate_gpu.cu:14-61), ragged sizes and a problem whose unknown count is not a multiple of the tile."""
    b = 3
    sizes = [(64, 32), (128, 64), (300, 128), (1024, 512), (70, 3)]
    pairs = [(_cloud(b, n, seed=n, kind="shell").to(DEV), _cloud(b, m, seed=n + m, kind="shell").to(DEV)) for n, m in sizes]
    res = ext.three_nn_weights_multi(pairs)
    assert len(res) == len(pairs)
    for (u, k), (idx, w) in zip(pairs, res):
        i1, w1 = ext.three_nn_weights(u, k)
        assert torch.equal(idx, i1) and torch.equal(w, w1)
        assert torch.equal(idx.cpu(), oracle.three_nn(u.cpu(), k.cpu())[1])
    with pytest.raises(Exception):
        ext.three_nn_weights_multi([])


@pytest.mark.parametrize("b,c,n,m", [(2, 3, 1024, 512), (2, 64, 512, 256), (1, 7, 100, 33), (1, 1, 5, 9)])
def test_gather_points_and_grad(ext, oracle, b, c, n, m):
    g = torch.Generator().manual_seed(c * n)
    pts = torch.randn(b, c, n, generator=g)
    idx = torch.randint(0, n, (b, m), generator=g, dtype=torch.int32)
    assert torch.equal(ext.gather_points(pts.to(DEV), idx.to(DEV)).cpu(), oracle.gather_points(pts, idx))
    go = torch.randn(b, c, m, generator=g)
    want = oracle.gather_points_grad(go, idx, n)
    got = ext.gather_points_grad(go.to(DEV), idx.to(DEV), n).cpu()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("b,c,n,npoint,nsample", [
    (4, 3, 1024, 512, 32), (2, 64, 512, 256, 16), (2, 128, 256, 128, 32), (2, 256, 128, 64, 32),
    (1, 5, 77, 13, 7), (1, 1, 4, 1, 1), (1, 9, 20000, 16, 8),
])
def test_group_points_and_grad(ext, oracle, b, c, n, npoint, nsample):
    g = torch.Generator().manual_seed(c + n + npoint)
    pts = torch.randn(b, c, n, generator=g)
    idx = torch.randint(0, n, (b, npoint, nsample), generator=g, dtype=torch.int32)
    # repeated indices (padded balls) accumulate in the gradient
    idx[:, :, nsample // 2:] = idx[:, :, :1]
    want = oracle.group_points(pts, idx)
    got = ext.group_points(pts.to(DEV), idx.to(DEV)).cpu()
    assert torch.equal(got, want)
    go = torch.randn(b, c, npoint, nsample, generator=g)
    gwant = oracle.group_points_grad(go, idx, n)
    ggot = ext.group_points_grad(go.to(DEV), idx.to(DEV), n).cpu()
    torch.testing.assert_close(ggot, gwant, rtol=1e-4, atol=1e-4)
    # equals a dense scatter-add in float64
    ref = torch.zeros(b, c, n, dtype=torch.float64)
    ref.scatter_add_(2, idx.long().reshape(b, 1, -1).expand(b, c, -1), go.double().reshape(b, c, -1))
    torch.testing.assert_close(ggot.double(), ref, rtol=1e-4, atol=1e-4)
    # the reference-shaped C entry (LDS atomics, no lists) stays a supported route: same sums
    from istnet_amd import _native
    god, idxd = go.to(DEV), idx.to(DEV)
    out = torch.empty(b, c, n, device=DEV)
    _native.check(_native.lib().istnet_pn2_group_points_grad(b, c, n, npoint, nsample, god.data_ptr(), idxd.data_ptr(),
                                                             out.data_ptr(), torch.cuda.current_stream().cuda_stream), "gg")
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-4, atol=1e-4)
    # the list route (default up to 4096 slots per cloud, on request up to 16384) is deterministic
    if npoint * nsample <= 16384:
        csr = ext.ball_csr(idxd, n)
        assert csr is not None
        a = ext.group_points_grad(god, idxd, n, csr=csr).cpu()
        torch.testing.assert_close(a.double(), ref, rtol=1e-4, atol=1e-4)
        assert torch.equal(ext.group_points_grad(god, idxd, n, csr=csr).cpu(), a)


@pytest.mark.parametrize("b,c,m,n", [(2, 512, 64, 128), (2, 256, 256, 512), (2, 256, 512, 1024), (1, 3, 4, 2),
                                     (1, 10, 20000, 50)])
def test_three_interpolate_and_grad(ext, oracle, b, c, m, n):
    g = torch.Generator().manual_seed(c + m + n)
    feats = torch.randn(b, c, m, generator=g)
    idx = torch.randint(0, m, (b, n, 3), generator=g, dtype=torch.int32)
    w = torch.rand(b, n, 3, generator=g)
    w = w / w.sum(dim=2, keepdim=True)
    want = oracle.three_interpolate(feats, idx, w)
    got = ext.three_interpolate(feats.to(DEV), idx.to(DEV), w.to(DEV)).cpu()
    assert torch.equal(got, want)  # same un-contracted f32 expression
    go = torch.randn(b, c, n, generator=g)
    gwant = oracle.three_interpolate_grad(go, idx, w, m)
    ggot = ext.three_interpolate_grad(go.to(DEV), idx.to(DEV), w.to(DEV), m).cpu()
    torch.testing.assert_close(ggot, gwant, rtol=1e-4, atol=1e-4)


def test_three_interpolate_reference_fixture(ext):
    """The one known answer the reference holds: pointnet2_test.py:25-30."""
    feats = torch.randn(1, 2, 4, generator=torch.Generator().manual_seed(1))
    idx = torch.tensor([[[0, 1, 2], [1, 2, 3]]], dtype=torch.int32)
    w = torch.tensor([[[1., 1, 1], [2, 2, 2]]])
    got = ext.three_interpolate(feats.to(DEV), idx.to(DEV), w.to(DEV)).cpu()
    want = torch.stack([feats[..., 0] + feats[..., 1] + feats[..., 2],
                        2 * (feats[..., 1] + feats[..., 2] + feats[..., 3])], dim=-1)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)


def test_cpu_tensors_rejected(ext):
    x = torch.rand(1, 8, 3)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.ball_query(x, x, 0.1, 4)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.furthest_point_sampling(x, 4)


def test_argument_checks(ext):
    x = torch.rand(1, 8, 3, device=DEV)
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.furthest_point_sampling(x.transpose(1, 2), 4)
    with pytest.raises(RuntimeError, match="float tensor"):
        ext.furthest_point_sampling(x.double(), 4)
    with pytest.raises(RuntimeError, match="int tensor"):
        ext.gather_points(x.transpose(1, 2).contiguous(), torch.zeros(1, 4, dtype=torch.int64, device=DEV))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ext.ball_query(x, x.cpu(), 0.1, 4)


def test_runs_on_current_stream(ext, oracle):
    xyz = _cloud(2, 256, seed=9)
    s = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(s):
        got = ext.furthest_point_sampling(xyz.to(DEV), 64)
    s.synchronize()
    assert torch.equal(got.cpu(), oracle.furthest_point_sampling(xyz, 64))


@pytest.mark.parametrize("n,m,kind", [(1024, 512, "shell"), (300, 77, "dup"), (4096, 64, "cube"), (1000, 333, "grid")])
def test_fps_gather_matches_fps_plus_gather(ext, oracle, n, m, kind):
    """istnet_pn2_fps_gather = furthest_point_sampling + gather of the picked coordinates, bit-exact."""
    xyz = _cloud(3, n, seed=n + m, kind=kind)
    idx_ref = oracle.furthest_point_sampling(xyz, m)
    idx, picked = ext.furthest_point_sampling_gather(xyz.to(DEV), m)
    assert torch.equal(idx.cpu(), idx_ref)
    expect = torch.gather(xyz, 1, idx_ref.long().unsqueeze(-1).expand(-1, -1, 3))
    assert torch.equal(picked.cpu(), expect)


def _first_tie_round(xyz, m):
    """CPU brute force in the kernels' arithmetic (IEEE f32, ((dx*dx + dy*dy) + dz*dz), no fusion): for every cloud the
    first round of furthest point sampling whose maximum is attained by more than one point (INT_MAX if none), taking
    the picks from ``picks`` (the oracle's, so the tie rule plays no part here)."""
    from oracle import pn2_oracle
    picks = pn2_oracle.furthest_point_sampling(xyz, m).long()
    b, n, _ = xyz.shape
    out = torch.full((b,), 2**31 - 1, dtype=torch.int64)
    for c in range(b):
        p = xyz[c]
        run = torch.full((n,), 1e10, dtype=torch.float32)
        for j in range(1, m):
            d = p - p[picks[c, j - 1]]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            run = torch.minimum(run, d2)
            if int((run == run.max()).sum()) > 1:
                out[c] = j
                break
    return out


@pytest.mark.parametrize("n,levels,kind,b", [
    (1024, (512, 256, 128, 64), "shell", 4),      # the encoder's chain (modules.py)
    (1024, (512, 256, 128, 64), "cube", 3), (1024, (512, 256, 128, 64), "dup", 3),
    (1024, (512, 256, 128, 64), "grid", 2), (2048, (1024, 512, 256, 128), "shell", 2),     # four waves per cloud
    (2048, (700, 300, 300, 17), "dup", 2), (700, (333, 120, 50, 50), "cube", 3), (300, (299, 150, 2, 1), "shell", 2),
])
def test_chained_fps_equals_level_by_level_sampling(ext, oracle, n, levels, kind, b):
    """istnet_pn2_fps_gather_chain: every level bit-identical to sampling it with the full scan (the oracle), whether a
    cloud took the prefix shortcut (parent run without an arg-max tie in the rounds the child needs) or the scan;
    the reported first tied round equals a brute-force count of the maxima; un-tied clouds DO take the shortcut."""
    xyz = _cloud(b, n, seed=n + len(kind), kind=kind)
    if kind in ("shell", "cube"):          # one cloud with duplicates inside an otherwise tie-free batch
        xyz[-1, n // 2:] = xyz[-1, :n - n // 2]
    cur_c, cur_g, tie = xyz, xyz.to(DEV), None
    shortcut = 0
    for li, m in enumerate(levels):
        nxt = levels[li + 1] if li + 1 < len(levels) else 0
        track = min(nxt, m)
        idx_ref = oracle.furthest_point_sampling(cur_c, m)
        idx, picked, tie_out = ext.furthest_point_sampling_chain(cur_g, m, tie_in=tie, track_rounds=track)
        assert torch.equal(idx.cpu(), idx_ref), f"level {li}"
        new_c = torch.gather(cur_c, 1, idx_ref.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        assert torch.equal(picked.cpu(), new_c), f"level {li}"
        took = None if tie is None else (tie.cpu().long() >= m) & (m <= cur_c.shape[1])
        if took is not None:
            shortcut += int(took.sum())
            assert torch.equal(tie_out.cpu()[took], tie.cpu()[took])            # shortcut: the parent's report passes on
            ar = torch.arange(m, dtype=torch.int32)
            assert all(torch.equal(idx.cpu()[c], ar) for c in range(b) if took[c])
        scanned = torch.ones(b, dtype=torch.bool) if took is None else ~took
        if scanned.any() and track > 1:
            want = _first_tie_round(cur_c[scanned], track).clamp(max=2**31 - 1)
            want = torch.where(want < track, want, torch.full_like(want, track if track < m else 2**31 - 1))
            assert torch.equal(tie_out.cpu().long()[scanned], want), f"level {li}: first tied round"
        cur_c, cur_g, tie = new_c, picked, tie_out
    if kind == "shell":
        assert shortcut >= (b - 1) * (len(levels) - 1) - 1      # tie-free clouds skip the scan on every later level


def test_index_ops_random_shapes_bit_exact(ext, oracle):
    """Seeded fuzz over small random shapes / distributions: FPS, ball query and three_nn stay bit-exact with the
    oracle for sizes that are not multiples of any tile, m > n, heavy ties and empty balls."""
    rng = np.random.default_rng(2024)
    kinds = ["cube", "shell", "dup", "grid"]
    for trial in range(40):
        b = int(rng.integers(1, 4))
        n = int(rng.integers(1, 700))
        m = int(rng.integers(1, 2 * n + 2)) if trial % 5 == 0 else int(rng.integers(1, n + 1))
        kind = kinds[trial % 4]
        xyz = _cloud(b, n, seed=1000 + trial, kind=kind)
        tag = f"trial {trial}: b={b} n={n} m={m} {kind}"
        want = oracle.furthest_point_sampling(xyz, m)
        got = ext.furthest_point_sampling(xyz.to(DEV), m).cpu()
        assert torch.equal(got, want), tag
        centers = torch.gather(xyz, 1, want.long().clamp_(0, n - 1).unsqueeze(-1).expand(b, m, 3)).contiguous()
        centers = centers + (torch.rand(b, m, 3, generator=torch.Generator().manual_seed(trial)) - 0.5) * (0.1 if trial % 3 == 0 else 0.0)
        radius = float(rng.choice([0.01, 0.05, 0.2, 0.5, 2.0]))
        nsample = int(rng.integers(1, 70))
        assert torch.equal(ext.ball_query(centers.to(DEV), xyz.to(DEV), radius, nsample).cpu(),
                           oracle.ball_query(centers, xyz, radius, nsample)), tag + f" r={radius} ns={nsample}"
        d_got, i_got = ext.three_nn(centers.to(DEV), xyz.to(DEV))
        d_want, i_want = oracle.three_nn(centers, xyz)
        assert torch.equal(i_got.cpu(), i_want), tag
        assert torch.equal(d_got.cpu().view(torch.int32), d_want.view(torch.int32)), tag


@pytest.mark.parametrize("conv", [1, 2])
def test_index_ops_bit_exact_under_fma_conventions(ext, oracle, conv):
    """The index-deciding distances are un-contracted by default; a reference build with nvcc's
    default -fmad=true may fuse them.  Under each of the two contracted forms (oracle.set_convention /
    istnet_pn2_set_tuning(1, c)) the HIP kernels stay bit-exact against the oracle -- including dist2 of three_nn,
    18 % of whose values change in the last bit -- on the clouds where tools/fma_flip_table.py finds flips
    (config-2 cube) and on tie-heavy inputs."""
    from istnet_amd import _native
    lib = _native.lib()
    assert lib.istnet_pn2_set_tuning(1, 3) != 0 and lib.istnet_pn2_set_tuning(0, 2000) != 0   # validated knobs
    g = torch.Generator().manual_seed(0)
    cube = torch.rand(32, 1024, 3, generator=g) * 0.2 - 0.1
    cases = [(cube - cube.mean(1, keepdim=True)).contiguous(), _cloud(4, 1024, 3, "shell"), _cloud(2, 1024, 5, "dup"),
             _cloud(2, 1000, 6, "grid"), _cloud(1, 2048, 7, "cube", 0.3), _cloud(1, 5000, 8, "cube")]
    base = {}
    for ci, xyz in enumerate(cases):
        base[ci] = oracle.furthest_point_sampling(xyz, min(512, xyz.shape[1] // 2))
    prev = oracle.set_convention(conv)
    assert lib.istnet_pn2_set_tuning(1, conv) == 0
    try:
        changed = 0
        for ci, xyz in enumerate(cases):
            b, n, _ = xyz.shape
            m = min(512, n // 2)
            fps = oracle.furthest_point_sampling(xyz, m)
            changed += int((fps != base[ci]).sum())
            assert torch.equal(ext.furthest_point_sampling(xyz.to(DEV), m).cpu(), fps)
            new_xyz = torch.gather(xyz, 1, fps.long().unsqueeze(-1).expand(b, m, 3)).contiguous()
            for radius, ns in ((0.02, 16), (0.04, 32), (0.25, 8)):
                want = oracle.ball_query(new_xyz, xyz, radius, ns)
                assert torch.equal(ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), radius, ns).cpu(), want)
            d_want, i_want = oracle.three_nn(xyz, new_xyz)
            d_got, i_got = ext.three_nn(xyz.to(DEV), new_xyz.to(DEV))
            assert torch.equal(i_got.cpu(), i_want) and torch.equal(d_got.cpu(), d_want)
        assert changed > 0    # the convention is really in effect: the cube cloud has FPS picks that depend on it
    finally:
        oracle.set_convention(prev)
        assert lib.istnet_pn2_set_tuning(1, 0) == 0


@pytest.mark.parametrize("b,n,m", [(4, 1024, 512), (2, 128, 64), (1, 77, 5), (1, 10, 3), (2, 33, 2)])
def test_three_nn_with_weights_in_one_launch(ext, b, n, m):
    """istnet_pn2_three_nn_weights: the indices of three_nn bit for bit, and the inverse-distance weights of
    PointnetFPModule (reference pointnet2_modules.py:185-188) as the five tensor ops compute them."""
    g = torch.Generator().manual_seed(n + m)
    unk = torch.rand(b, n, 3, generator=g).to(DEV)
    kn = torch.rand(b, m, 3, generator=g).to(DEV)
    kn[:, 0] = unk[:, 0]                                   # a zero distance: weight 1e8 / (1e8 + ...) after the epsilon
    d2, idx = ext.three_nn(unk, kn)
    idx2, w = ext.three_nn_weights(unk, kn)
    assert torch.equal(idx2, idx)
    inv = 1.0 / (torch.sqrt(d2) + 1e-8)
    want = inv / inv.sum(dim=2, keepdim=True)
    torch.testing.assert_close(w, want, rtol=2e-6, atol=1e-7, equal_nan=True)
    if m >= 3:
        torch.testing.assert_close(w.sum(dim=2), torch.ones(b, n, device=DEV), rtol=1e-6, atol=1e-6)

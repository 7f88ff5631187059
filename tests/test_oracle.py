"""CPU tests of the oracle itself: known answers derived from the reference .cu semantics
(SURVEY.md 8c), the single reference-held fixture (pointnet2_test.py:25-30) and the committed
golden vectors."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_opt_n_threads(oracle):
    # cuda_utils.h:18-22
    for n, want in [(1, 1), (2, 2), (3, 2), (64, 64), (65, 64), (511, 256), (512, 512), (1000, 512), (1024, 512),
                    (5000, 512)]:
        assert oracle.opt_n_threads(n) == want


def test_three_interpolate_reference_fixture(oracle):
    feats = torch.randn(1, 2, 4, generator=torch.Generator().manual_seed(1))
    idx = torch.tensor([[[0, 1, 2], [1, 2, 3]]], dtype=torch.int32)
    w = torch.tensor([[[1., 1, 1], [2, 2, 2]]])
    out = oracle.three_interpolate(feats, idx, w)
    want = torch.stack([feats[..., 0] + feats[..., 1] + feats[..., 2],
                        2 * (feats[..., 1] + feats[..., 2] + feats[..., 3])], dim=-1)
    torch.testing.assert_close(out, want, rtol=1e-6, atol=1e-6)
    # gradient = transpose of the same linear map
    go = torch.ones(1, 2, 2)
    grad = oracle.three_interpolate_grad(go, idx, w, 4)
    torch.testing.assert_close(grad[0, 0], torch.tensor([1., 3, 3, 2]))


def test_ball_query_semantics(oracle):
    xyz = torch.tensor([[[0., 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [0.1, 0, 0]]])
    new_xyz = torch.tensor([[[10., 10, 10], [1, 0, 0], [0, 0, 0]]])
    idx = oracle.ball_query(new_xyz, xyz, 0.5, 4)
    assert idx[0, 0].tolist() == [0, 0, 0, 0]          # no neighbour -> zeros (ball_query.cpp:24-26)
    assert idx[0, 1].tolist() == [1, 1, 1, 1]          # single hit fills the row (:39-43)
    assert idx[0, 2].tolist() == [0, 4, 0, 0]          # hits in index order, padded with the first
    idx = oracle.ball_query(new_xyz, xyz, 1.0, 2)       # d2 == r2 is excluded (strict <, :38)
    assert idx[0, 1].tolist() == [1, 4]
    idx = oracle.ball_query(new_xyz, xyz, 100.0, 3)     # more hits than nsample -> first nsample
    assert idx[0, 0].tolist() == [0, 1, 2]


def test_fps_semantics(oracle):
    g = torch.Generator().manual_seed(0)
    xyz = torch.rand(3, 200, 3, generator=g)
    idx = oracle.furthest_point_sampling(xyz, 50)
    assert (idx[:, 0] == 0).all()
    for b in range(3):
        assert len(set(idx[b].tolist())) == 50          # distinct points: no repeats
        # greedy property: each pick maximises the distance to the set so far
        d = torch.full((200,), 1e10)
        for j in range(1, 50):
            last = xyz[b, idx[b, j - 1]]
            d = torch.minimum(d, ((xyz[b] - last) ** 2).sum(-1))
            assert torch.isclose(d[idx[b, j]], d.max(), rtol=1e-6)


def _bitrev(v, bits):
    r = 0
    for i in range(bits):
        r |= ((v >> i) & 1) << (bits - 1 - i)
    return r


def test_fps_tie_break_is_block_tree_order(oracle):
    """sampling_gpu.cu:64-70,120-173: with equal distances the winner is the thread whose slot has
    the smallest BIT-REVERSED index (tree keeps the lower slot on ties), then the lowest k."""
    # point 0 at the origin, all others on a sphere of equal (exactly representable) distance
    n = 16
    xyz = torch.zeros(1, n, 3)
    corners = torch.tensor([[1., 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])
    for k in range(1, n):
        xyz[0, k] = corners[(k - 1) % 6]
    idx = oracle.furthest_point_sampling(xyz, 2)
    # all k >= 1 tie at d = 1; bs = 16 threads, one point each: winner = argmin bitrev4(k), k>=1 -> k=8
    assert idx[0, 1].item() == min(range(1, n), key=lambda k: _bitrev(k, 4)) == 8
    # n = 24: bs = 16, threads 0..7 own two points (k, k+16).  Thread 0 holds k=0 (d=0) and k=16
    # (d=1): its maximum is k=16, and slot 0 has the smallest bit-reversed index -> winner 16.
    n = 24
    xyz = torch.zeros(1, n, 3)
    for k in range(1, n):
        xyz[0, k] = corners[(k - 1) % 6]
    idx = oracle.furthest_point_sampling(xyz, 2)
    assert idx[0, 1].item() == 16


def test_three_nn_semantics(oracle):
    unknown = torch.tensor([[[0., 0, 0]]])
    known = torch.tensor([[[1., 0, 0], [0, 1, 0], [0, 0, 1], [0.5, 0, 0]]])
    d2, idx = oracle.three_nn(unknown, known)
    assert idx[0, 0].tolist() == [3, 0, 1]               # ascending, earliest index wins ties
    assert d2[0, 0].tolist() == [0.25, 1.0, 1.0]
    d2, idx = oracle.three_nn(unknown, known[:, :2])      # m < 3 -> idx 0 and +inf
    assert idx[0, 0].tolist() == [0, 1, 0]
    assert d2[0, 0, 2].item() == float("inf")


def test_group_and_gather_grads_are_scatter_adds(oracle):
    g = torch.Generator().manual_seed(3)
    idx = torch.randint(0, 10, (2, 4, 5), generator=g, dtype=torch.int32)
    go = torch.randn(2, 3, 4, 5, generator=g)
    ref = torch.zeros(2, 3, 10).scatter_add_(2, idx.long().reshape(2, 1, -1).expand(2, 3, -1), go.reshape(2, 3, -1))
    torch.testing.assert_close(oracle.group_points_grad(go, idx, 10), ref)
    idx1 = torch.tensor([[1, 1, 3]], dtype=torch.int32)
    got = oracle.gather_points_grad(torch.ones(1, 2, 3), idx1, 5)
    assert got[0, 0].tolist() == [0, 2, 0, 1, 0]


def test_oracle_reproduces_golden_config1(oracle):
    z = np.load(os.path.join(GOLD, "config1_sa_grouping.npz"))
    xyz = torch.from_numpy(z["xyz"])
    fps = oracle.furthest_point_sampling(xyz, 512)
    assert np.array_equal(fps.numpy(), z["fps_idx"].astype(np.int32))
    new_xyz = torch.from_numpy(z["new_xyz"])
    bq = oracle.ball_query(new_xyz, xyz, 0.2, 32)
    assert np.array_equal(bq.numpy(), z["ball_idx"].astype(np.int32))


def test_oracle_reproduces_golden_encoder_indices(oracle):
    z = np.load(os.path.join(GOLD, "encoder_b2.npz"))
    xyz = torch.from_numpy(z["pts"])
    npoints = [512, 256, 128, 64]
    radii = [[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]]
    levels = [xyz]
    for lvl in range(4):
        fps = oracle.furthest_point_sampling(levels[-1], npoints[lvl])
        assert np.array_equal(fps.numpy(), z[f"furthest_point_sampling_{lvl}"].astype(np.int32))
        new_xyz = torch.gather(levels[-1], 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        for s, (r, ns) in enumerate(zip(radii[lvl], (16, 32))):
            bq = oracle.ball_query(new_xyz, levels[-1], r, ns)
            assert np.array_equal(bq.numpy(), z[f"ball_query_{2 * lvl + s}"].astype(np.int32))
        levels.append(new_xyz)
    for i, lvl in enumerate(range(3, -1, -1)):            # FP runs coarse -> fine
        d2, idx = oracle.three_nn(levels[lvl], levels[lvl + 1])
        assert np.array_equal(idx.numpy(), z[f"three_nn_idx_{i}"].astype(np.int32))
        assert np.array_equal(d2.numpy(), z[f"three_nn_dist2_{i}"])


@pytest.mark.parametrize("conv", [0, 1, 2])
def test_oracle_reproduces_the_convention_goldens(conv):
    """tests/golden/index_conventions.npz (reference Python over the oracle, per arithmetic convention of the squared
    distances; make_golden_conventions.py): the oracle's stand-alone ops give the stored config-1 indices and the level-1
    tensors of the cube clouds again, and the three conventions really differ on the stored clouds."""
    import os
    import numpy as np
    import torch
    from oracle import pn2_oracle
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "index_conventions.npz"))
    prev = pn2_oracle.set_convention(conv)
    try:
        xyz1 = torch.from_numpy(z["xyz_config1"])
        fps = pn2_oracle.furthest_point_sampling(xyz1, 512)
        assert np.array_equal(fps.numpy(), z[f"c{conv}_config1_fps"].astype(np.int32))
        new_xyz = torch.gather(xyz1, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        assert np.array_equal(pn2_oracle.ball_query(new_xyz, xyz1, 0.2, 32).numpy(), z[f"c{conv}_config1_ball"].astype(np.int32))
        pts = torch.from_numpy(z["pts_cube"])
        fps = pn2_oracle.furthest_point_sampling(pts, 512)
        assert np.array_equal(fps.numpy(), z[f"c{conv}_furthest_point_sampling_0"].astype(np.int32))
        new_xyz = torch.gather(pts, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        for i, (r, s) in enumerate(((0.01, 16), (0.02, 32))):
            assert np.array_equal(pn2_oracle.ball_query(new_xyz, pts, r, s).numpy(), z[f"c{conv}_ball_query_{i}"].astype(np.int32))
    finally:
        pn2_oracle.set_convention(prev)
    if conv:
        keys = [k for k in z.files if k.startswith("c0_") and z[k].dtype == np.int16]
        assert sum(int((z[k] != z[f"c{conv}_" + k[3:]]).sum()) for k in keys) == 138

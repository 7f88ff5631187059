"""istnet_amd.align (batched Umeyama + RANSAC) against golden vectors produced by the reference's own utils/align.py
(tests/golden/make_golden_align.py): same transforms for the same random draws, None-cases flagged, float64."""
import os

import numpy as np
import pytest
import torch

import istnet_amd  # noqa: F401
from istnet_amd import align

GOLD = os.path.join(os.path.dirname(__file__), "golden", "align.npz")


def _check(device):
    z = np.load(GOLD)
    src = torch.from_numpy(z["source"]).to(device)
    tgt = torch.from_numpy(z["target"]).to(device)
    n = src.shape[1]
    scale, rot, trans, tf, ok, info = align.estimate_similarity_transform(src, tgt, seeds=z["seeds"].tolist())
    assert tf.dtype == torch.float64 and tf.device.type == torch.device(device).type
    assert ok.cpu().numpy().tolist() == z["ok"].tolist()            # the reference returned None exactly there
    good = z["ok"]
    assert good.sum() >= 8 and (~good).sum() >= 3                   # both outcomes are covered
    np.testing.assert_allclose(tf.cpu().numpy()[good], z["transform"][good], rtol=1e-9, atol=1e-11)
    assert np.isnan(tf.cpu().numpy()[~good]).all()
    # consistency of the parts with the assembled transform
    np.testing.assert_allclose((scale.view(-1, 1, 1) * rot).cpu().numpy()[good], z["transform"][good][:, :3, :3], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(trans.cpu().numpy()[good], z["transform"][good][:, :3, 3], rtol=1e-9, atol=1e-11)
    # explicit draws = what the seeds expand to; a single instance goes through the same path
    ridx = align.draw_indices(n, z["seeds"].tolist())
    tf2 = align.estimate_similarity_transform(src, tgt, rand_idx=ridx)[3]
    assert torch.equal(torch.nan_to_num(tf2), torch.nan_to_num(tf))
    one = align.estimate_similarity_transform(src[0], tgt[0], rand_idx=ridx[0])
    assert bool(one[4]) and torch.equal(one[3], tf[0])
    # plain Umeyama, and its masked form
    _, _, _, tfu = align.umeyama(src[:4, :64], tgt[:4, :64])
    np.testing.assert_allclose(tfu.cpu().numpy(), z["umeyama64"], rtol=1e-10, atol=1e-12)
    mask = torch.zeros(4, n, dtype=torch.bool, device=device)
    mask[:, :64] = True
    np.testing.assert_allclose(align.umeyama(src[:4], tgt[:4], mask)[3].cpu().numpy(), z["umeyama64"], rtol=1e-10, atol=1e-12)
    # the early exit is taken where the data are clean, and every iteration runs where they are not
    runs = info["iterations_run"].cpu().numpy()
    assert runs.min() < 10 and runs.max() == align.MAX_ITER


def test_align_matches_reference_cpu():
    _check("cpu")


@pytest.mark.gpu
def test_align_matches_reference_gpu():
    _check("cuda:0")


def test_align_recovers_a_known_transform_and_rejects_bad_input():
    g = torch.Generator().manual_seed(0)
    src = torch.rand(3, 500, 3, generator=g, dtype=torch.float64) - 0.5
    q = torch.linalg.qr(torch.randn(3, 3, 3, generator=g, dtype=torch.float64))[0]
    q = q * torch.sign(torch.linalg.det(q)).view(-1, 1, 1)
    s = torch.tensor([0.2, 1.0, 3.5], dtype=torch.float64)
    t = torch.randn(3, 3, generator=g, dtype=torch.float64)
    tgt = s.view(-1, 1, 1) * (src @ q.transpose(1, 2)) + t.unsqueeze(1)
    tgt[:, :100] += 50.0 + torch.randn(3, 100, 3, generator=g, dtype=torch.float64)   # 20 % gross outliers
    scale, rot, trans, tf, ok, _ = align.estimate_similarity_transform(src, tgt)
    assert bool(ok.all())
    torch.testing.assert_close(scale, s, rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(rot, q, rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(trans, t, rtol=1e-9, atol=1e-9)
    with pytest.raises(AssertionError):
        align.estimate_similarity_transform(src, tgt[:, :10])
    with pytest.raises(RuntimeError):
        align.umeyama(src * float("nan"), tgt)

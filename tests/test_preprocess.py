"""Input preparation: numpy oracle known answers on CPU, HIP kernel vs oracle bit-exact on GPU."""
import numpy as np
import pytest
import torch

import istnet_amd  # noqa: F401
from oracle import preproc_oracle

K = (591.0125, 590.16775, 322.525, 244.11084)


def _scene(seed, h=480, w=640, kind="u16"):
    rng = np.random.default_rng(seed)
    depth = rng.integers(300, 3000, (h, w)).astype(np.uint16)
    depth[rng.random((h, w)) < 0.05] = 0
    if kind == "f32":
        depth = (depth.astype(np.float32) / 1000.0 * 1 + rng.random((h, w)).astype(np.float32) * 1e-3) / 1 * 1000.0
        assert depth.dtype == np.float32
    return rng, depth


def _instances(rng, count, n, h=480, w=640):
    boxes, chooses = [], []
    for side in rng.choice([40, 80, 120, 160, 200, 280, 440], count):
        rmin, cmin = int(rng.integers(0, h - side + 1)), int(rng.integers(0, w - side + 1))
        boxes.append((rmin, rmin + int(side), cmin, cmin + int(side)))
        chooses.append(rng.integers(0, side * side, n))
    return np.array(boxes), np.stack(chooses)


def test_oracle_known_answers():
    depth = np.zeros((480, 640), dtype=np.uint16)
    depth[100, 200] = 1500
    depth[101, 203] = 800
    bbox = (90, 130, 190, 230)                       # 40 x 40 crop
    choose = np.array([10 * 40 + 10, 11 * 40 + 13, 0])
    pts, out = preproc_oracle.backproject_choose(depth, bbox, choose, K)
    assert pts.dtype == np.float32 and out.dtype == np.int64
    np.testing.assert_array_equal(pts[0], np.float32([(200 - K[2]) * 1.5 / K[0], (100 - K[3]) * 1.5 / K[1], 1.5]))
    np.testing.assert_array_equal(pts[1], np.float32([(203 - K[2]) * 0.8 / K[0], (101 - K[3]) * 0.8 / K[1], 0.8]))
    np.testing.assert_array_equal(pts[2], np.zeros(3, np.float32))
    # 40 -> 192: ratio 4.8; (10,10) -> (48,48), (11,13) -> (52,62)
    np.testing.assert_array_equal(out, [48 * 192 + 48, 52 * 192 + 62, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["u16", "f32"])
@pytest.mark.parametrize("shared", [True, False])
def test_backproject_matches_oracle_bit_exact(kind, shared):
    from istnet_amd import preprocess
    count, n = 5, 1024
    rng, depth = _scene(7, kind=kind)
    images = [depth] if shared else [_scene(20 + i, kind=kind)[1] for i in range(count)]
    boxes, choose = _instances(rng, count, n)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(np.stack(images) if not shared else images[0])
    if kind == "u16":
        d = d.view(torch.int16) if d.dtype != torch.uint16 else d
    pts, out = preprocess.backproject_choose(d.to(dev), torch.from_numpy(boxes), torch.from_numpy(choose), K)
    for i in range(count):
        want_pts, want_out = preproc_oracle.backproject_choose(images[0 if shared else i], tuple(boxes[i]), choose[i], K)
        np.testing.assert_array_equal(pts[i].cpu().numpy().view(np.uint32), want_pts.view(np.uint32))
        np.testing.assert_array_equal(out[i].cpu().numpy(), want_out)
    assert int(out.max()) < 192 * 192 and int(out.min()) >= 0


@pytest.mark.gpu
def test_backproject_edge_cases_and_errors():
    from istnet_amd import preprocess
    dev = torch.device("cuda:0")
    depth = torch.zeros(480, 640, dtype=torch.float32, device=dev)
    pts, out = preprocess.backproject_choose(depth, torch.zeros(0, 4, dtype=torch.int32), torch.zeros(0, 16, dtype=torch.int64))
    assert pts.shape == (0, 16, 3) and out.shape == (0, 16)
    pts, out = preprocess.backproject_choose(depth, torch.tensor([[0, 40, 0, 40]]), torch.zeros(1, 0, dtype=torch.int64))
    assert pts.shape == (1, 0, 3)
    with pytest.raises(RuntimeError):
        preprocess.backproject_choose(depth.cpu(), torch.tensor([[0, 40, 0, 40]]), torch.zeros(1, 4, dtype=torch.int64))
    with pytest.raises(TypeError):
        preprocess.backproject_choose(depth.double(), torch.tensor([[0, 40, 0, 40]]), torch.zeros(1, 4, dtype=torch.int64))
    with pytest.raises(ValueError):
        preprocess.backproject_choose(depth, torch.tensor([[0, 40, 0, 40]]), torch.zeros(2, 4, dtype=torch.int64))

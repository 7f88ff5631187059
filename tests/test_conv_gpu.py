"""include/istnet_conv.h (channels-last float32 convolution as an implicit GEMM on the fp32 matrix cores: the RGB trunk's
3x3 / 1x1 layers, reference model/resnet.py:18-67,109-202) against a float64 evaluation of the same convolution: forward,
input gradient, weight gradient, at the trunk's layer shapes and at ragged sizes; determinism; the autograd node."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0") if torch.cuda.is_available() else None

# (cin, cout, k, stride, h, w): every convolution shape of the ResNet-18 trunk at 1/4 .. 1/8 resolution, then ragged sizes
# (row counts that are not multiples of the 128-pixel tile, pixel counts that are not multiples of the 32-pixel chunk)
LAYERS = [(64, 64, 3, 1, 48, 48), (64, 128, 3, 2, 48, 48), (128, 128, 3, 1, 24, 24), (64, 128, 1, 2, 48, 48),
          (128, 256, 3, 1, 24, 24), (256, 256, 3, 1, 24, 24), (128, 256, 1, 1, 24, 24), (256, 512, 3, 1, 24, 24),
          (512, 512, 3, 1, 24, 24), (256, 512, 1, 1, 24, 24), (64, 64, 3, 1, 10, 14), (128, 64, 3, 2, 9, 11),
          (64, 192, 1, 1, 5, 7), (64, 64, 5, 1, 12, 12)]


def _st():
    return torch.cuda.current_stream().cuda_stream


def _ws(lib, bd, args):
    n = lib.istnet_conv_workspace_floats(bd, *args)
    assert n >= 0
    return torch.empty(max(n, 1), device=DEV)


@pytest.mark.parametrize("cin,cout,k,s,h,w", LAYERS)
@pytest.mark.parametrize("b", [2, 5])
def test_conv_products_match_float64(cin, cout, k, s, h, w, b):
    from istnet_amd import _native
    lib = _native.lib()
    pad = k // 2
    assert lib.istnet_conv_supported(cin, cout, k, k, s, pad) == 1
    g = torch.Generator().manual_seed(cin + cout + k + h)
    x = torch.randn(b, cin, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    wgt = (torch.randn(cout, cin, k, k, generator=g) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last)
    oh, ow = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    dy = torch.randn(b, cout, oh, ow, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    xd, wd = x.double().contiguous().requires_grad_(True), wgt.double().contiguous().requires_grad_(True)
    yd = torch.nn.functional.conv2d(xd, wd, None, s, pad)
    yd.backward(dy.double())
    args = (b, h, w, cin, cout, k, k, s, pad)
    rel = lambda a, r: float((a.double() - r).abs().max() / (r.abs().max() + 1e-30))
    y = torch.empty(b, cout, oh, ow, device=DEV).contiguous(memory_format=torch.channels_last)
    assert lib.istnet_conv_forward(*args, x.data_ptr(), wgt.data_ptr(), y.data_ptr(), _ws(lib, 0, args).data_ptr(), _st()) == 0
    assert rel(y, yd.detach()) < 2e-6
    dx = torch.empty_like(x)
    assert lib.istnet_conv_backward_data(*args, dy.data_ptr(), wgt.data_ptr(), dx.data_ptr(), _ws(lib, 1, args).data_ptr(), _st()) == 0
    assert rel(dx, xd.grad) < 2e-6
    splits = lib.istnet_conv_wrw_splits(*args)
    if (oh * ow) % 32 == 0:
        assert splits >= 1
        part = torch.empty(splits, wgt.numel(), device=DEV)
        dw = torch.empty_like(wgt)
        assert lib.istnet_conv_backward_weights(*args, x.data_ptr(), dy.data_ptr(), part.data_ptr(), dw.data_ptr(), _st()) == 0
        assert rel(dw, wd.grad) < 2e-6
        dw2 = torch.empty_like(wgt)
        assert lib.istnet_conv_backward_weights(*args, x.data_ptr(), dy.data_ptr(), part.data_ptr(), dw2.data_ptr(), _st()) == 0
        assert torch.equal(dw, dw2)                          # split-K partials are summed in a fixed order
    else:
        assert splits == 0                                   # the host keeps the framework's product for such maps
    y2 = torch.empty_like(y)
    assert lib.istnet_conv_forward(*args, x.data_ptr(), wgt.data_ptr(), y2.data_ptr(), _ws(lib, 0, args).data_ptr(), _st()) == 0
    assert torch.equal(y, y2)


def test_conv_full_batch_layer_with_split_rounds():
    """The training batch's largest layer (B = 32, 24 x 24, 512 -> 512): 576 output tiles = one full round of the chip's
    workgroup slots unsplit + 64 tiles with K split eight ways through the work space; against the framework's convolution."""
    from istnet_amd import _native
    lib = _native.lib()
    b, c, h = 32, 512, 24
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, c, h, h, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    wgt = (torch.randn(c, c, 3, 3, generator=g) * 0.02).to(DEV).contiguous(memory_format=torch.channels_last)
    args = (b, h, h, c, c, 3, 3, 1, 1)
    assert lib.istnet_conv_workspace_floats(0, *args) > 0
    y = torch.empty_like(x)
    assert lib.istnet_conv_forward(*args, x.data_ptr(), wgt.data_ptr(), y.data_ptr(), _ws(lib, 0, args).data_ptr(), _st()) == 0
    ref = torch.nn.functional.conv2d(x, wgt, None, 1, 1)
    assert float((y - ref).abs().max() / ref.abs().max()) < 1e-5
    dx = torch.empty_like(x)
    assert lib.istnet_conv_backward_data(*args, ref.data_ptr(), wgt.data_ptr(), dx.data_ptr(), _ws(lib, 1, args).data_ptr(), _st()) == 0
    dref = torch.ops.aten.convolution_backward(ref, x, wgt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    assert float((dx - dref).abs().max() / dref.abs().max()) < 1e-5


def test_conv_rejects_what_it_does_not_cover():
    from istnet_amd import _native
    lib = _native.lib()
    assert lib.istnet_conv_supported(3, 64, 7, 7, 2, 3) == 0          # the stem: 3 input channels
    assert lib.istnet_conv_supported(64, 48, 3, 3, 1, 1) == 0
    assert lib.istnet_conv_supported(64, 64, 3, 3, 3, 1) == 0
    x = torch.zeros(1, device=DEV)
    assert lib.istnet_conv_forward(1, 8, 8, 3, 64, 7, 7, 2, 3, x.data_ptr(), x.data_ptr(), x.data_ptr(), None, _st()) != 0


def test_basic_block_with_native_convolutions():
    """BasicBlock (reference model/resnet.py:36-67) with its convolutions on include/istnet_conv.h against the same block on
    the framework's convolutions: output and every gradient (the 1x1 downsample layer takes the native weight gradient)."""
    from istnet_amd import rgb_branch
    torch.manual_seed(4)
    ds = torch.nn.Sequential(torch.nn.Conv2d(64, 128, 1, stride=2, bias=False), torch.nn.BatchNorm2d(128))
    blk = rgb_branch.BasicBlock(64, 128, stride=2, downsample=ds).to(DEV).to(memory_format=torch.channels_last).train()
    blk2 = rgb_branch.BasicBlock(128, 128).to(DEV).to(memory_format=torch.channels_last).train()
    x = torch.randn(4, 64, 16, 16, device=DEV).contiguous(memory_format=torch.channels_last)
    wgt = torch.randn(4, 128, 8, 8, device=DEV)

    def run(native):
        old = rgb_branch.USE_NATIVE_TRUNK_CONV
        rgb_branch.USE_NATIVE_TRUNK_CONV = native
        try:
            for m in (blk, blk2):
                m.zero_grad(set_to_none=True)
            xx = x.clone(memory_format=torch.preserve_format).requires_grad_(True)
            out = blk2(blk(xx))
            (out * wgt).sum().backward()
            grads = {n: p.grad.clone() for m in (blk, blk2) for n, p in m.named_parameters()}
            return out.detach(), xx.grad, grads
        finally:
            rgb_branch.USE_NATIVE_TRUNK_CONV = old

    a, r = run(True), run(False)
    rel = lambda u, v: float((u - v).abs().max() / (v.abs().max() + 1e-30))
    assert rel(a[0], r[0]) < 1e-4 and rel(a[1], r[1]) < 1e-3
    for name in a[2]:
        assert rel(a[2][name], r[2][name]) < 1e-3, name


@pytest.mark.parametrize("cin,cout,k,s,h,w", LAYERS)
@pytest.mark.parametrize("variant", [1, 2])
def test_split_precision_kernels_hold_the_fp32_error(cin, cout, k, s, h, w, variant):
    """Opt-in split precision (istnet_conv_set_tuning(1, v), include/istnet_conv.h): every fp32 operand as three exact bf16
    terms, six v_mfma_f32_32x32x16_bf16 products, fp32 accumulation.  The acceptance gate of the experiment, per shape and for
    forward AND backward-data (stride 1; stride 2 keeps the fp32 path): the error against a float64 convolution is at most 2x
    the exact-fp32 MFMA kernel's own error, in the maximum and in the rms (measured: 0.8-1.0x, profiles/r05_split_precision_conv.txt)."""
    from istnet_amd import _native
    lib = _native.lib()
    pad = k // 2
    b = 3
    g = torch.Generator().manual_seed(7 * cin + cout + k + h + variant)
    x = (torch.randn(b, cin, h, w, generator=g).abs() * torch.rand(b, cin, 1, 1, generator=g) * 3).to(DEV).contiguous(memory_format=torch.channels_last)
    wgt = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    oh, ow = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    dy = torch.randn(b, cout, oh, ow, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    y64 = torch.nn.functional.conv2d(x.double(), wgt.double(), None, s, pad)
    dx64 = torch.nn.grad.conv2d_input(x.shape, wgt.double(), dy.double(), s, pad)
    args = (b, h, w, cin, cout, k, k, s, pad)
    errs = {}
    try:
        for mode in (0, variant):
            assert lib.istnet_conv_set_tuning(1, mode) == 0 and lib.istnet_conv_get_tuning(1) == mode
            y = torch.empty((b, cout, oh, ow), device=DEV).contiguous(memory_format=torch.channels_last)
            assert lib.istnet_conv_forward(*args, x.data_ptr(), wgt.data_ptr(), y.data_ptr(), _ws(lib, 0, args).data_ptr(), _st()) == 0
            dx = torch.empty_like(x)
            if s == 1:
                assert lib.istnet_conv_backward_data(*args, dy.data_ptr(), wgt.data_ptr(), dx.data_ptr(), _ws(lib, 1, args).data_ptr(), _st()) == 0
            torch.cuda.synchronize()
            e = [(y.double() - y64).abs().max() / y64.abs().max(), (y.double() - y64).pow(2).mean().sqrt()]
            if s == 1:
                e += [(dx.double() - dx64).abs().max() / dx64.abs().max(), (dx.double() - dx64).pow(2).mean().sqrt()]
            errs[mode] = [float(v) for v in e]
    finally:
        assert lib.istnet_conv_set_tuning(1, 0) == 0
    assert lib.istnet_conv_set_tuning(1, 99) != 0
    for e_split, e_fp32 in zip(errs[variant], errs[0]):
        assert e_split <= 2.0 * e_fp32 + 1e-12, (errs[variant], errs[0])
    assert errs[variant][0] < 1e-5


def test_split_precision_serves_inference_through_the_native_forward():
    """With split precision on, the native forward is also taken without gradients (eval / torch.no_grad: the exact-fp32 native
    forward stays off there, rgb_branch._native_conv_ok); the eval-mode block agrees with the framework's convolutions to
    fp32 rounding, and turning the mode off restores the framework path."""
    from istnet_amd import rgb_branch
    torch.manual_seed(5)
    blk = rgb_branch.BasicBlock(128, 128).to(DEV).to(memory_format=torch.channels_last).eval()
    x = torch.randn(4, 128, 24, 24, device=DEV).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert not rgb_branch._native_conv_ok(blk.conv1, x)
        ref = blk(x)
        prev = rgb_branch.set_split_precision(True)
        try:
            assert rgb_branch._native_conv_ok(blk.conv1, x)
            got = blk(x)
        finally:
            rgb_branch.set_split_precision(prev)
        assert not rgb_branch._native_conv_ok(blk.conv1, x)
    y64 = blk.double()(x.double())
    e_ref, e_got = float((ref.double() - y64).abs().max()), float((got.double() - y64).abs().max())
    assert e_got <= 2.0 * e_ref + 1e-12 and e_got < 1e-4 * float(y64.abs().max()), (e_got, e_ref)

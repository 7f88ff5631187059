"""RGB branch of IST-Net: ResNet-18 trunk (output stride 8) + pyramid pooling + 3 x (2x upsample, 3x3 conv).

SURVEY.md 8(f) rank 1 -- outside the point-cloud hot path: dense 2-D convolutions that stay on
PyTorch-ROCm / MIOpen.  This file only restates the module tree of the reference
(model/modules.py:10-81,225-241 and model/resnet.py:31-60,109-214) with identical child names, so
that a reference checkpoint (`rgb_cam_extractor.model.feats.layer1.0.conv1.weight`, ...) loads
unchanged and the full IST-Net (`ist_net.IST_Net(rgb_extractor=ModifiedResnet())`) can be trained
and benchmarked end to end.  rgb (B,3,H,W) -> (B,128,H,W).

Reference quirks kept on purpose: the `dilation` arguments of layer3 / layer4 are ignored
(resnet.py:153-180 only dilates once the *output_stride* is reached, and that is 32), so both run at
stride 1, dilation 1 on the stride-8 map; `avgpool` / `fc` exist (state-dict keys) but are unused;
the pyramid priors are upsampled with align_corners=False, the decoder with align_corners=True.
No weights are downloaded (the reference fetches resnet18-5c106cde.pth, resnet.py:205-214).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


USE_NATIVE_DECODER_BACKWARD = True   # False: the framework's own backward of PReLU / bilinear upsample


class _PReLUFn(torch.autograd.Function):
    """nn.PReLU() with one slope: the framework's forward, include/istnet_rgb.h's backward (one streaming pass, partial
    sums of the slope gradient in a fixed order).  The framework's backward kernel walks the tensor with per-element
    stride arithmetic and ran at 0.35 TB/s on the decoder's channels-last maps -- 24 % of the branch's training step
    (profiles/r02_rgb_branch_breakdown.txt)."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return F.prelu(x, weight)

    @staticmethod
    def backward(ctx, dy):
        from . import _native
        x, weight = ctx.saved_tensors
        fmt = torch.channels_last if (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
                                      and not x.is_contiguous()) else torch.contiguous_format
        x = x.contiguous(memory_format=fmt)
        dy = dy.contiguous(memory_format=fmt)
        dx = torch.empty_like(x, memory_format=fmt)
        lib = _native.lib()
        n = x.numel()
        part = torch.empty(lib.istnet_prelu_bwd_parts(n), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _native.check(lib.istnet_prelu_bwd(n, x.data_ptr(), dy.data_ptr(), weight.data_ptr(), dx.data_ptr(),
                                               part.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream),
                          "prelu_bwd")
        return dx, part.sum().reshape(weight.shape)


class PReLU(nn.PReLU):
    """nn.PReLU (same parameter, same state-dict key) with the native backward on the GPU."""

    def forward(self, x):
        if (USE_NATIVE_DECODER_BACKWARD and x.is_cuda and x.dtype == torch.float32 and self.weight.numel() == 1
                and torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)):
            return _PReLUFn.apply(x, self.weight)
        return super().forward(x)


class _UpsampleAlignedFn(torch.autograd.Function):
    """Bilinear upsample with align_corners=True on a channels-last map: the framework's forward, a gather-form backward
    (every input pixel sums its weighted output pixels: no atomics, so the step stays bit-reproducible)."""

    @staticmethod
    def forward(ctx, x, size):
        ctx.in_shape = tuple(x.shape)
        return F.interpolate(x, size=size, mode="bilinear", align_corners=True)

    @staticmethod
    def backward(ctx, dy):
        from . import _native
        b, c, hin, win = ctx.in_shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty(ctx.in_shape, dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        with torch.cuda.device(dy.device):
            _native.check(_native.lib().istnet_upsample_bilinear_ac_bwd_nhwc(
                b, c, hin, win, dy.shape[2], dy.shape[3], dy.data_ptr(), dx.data_ptr(),
                torch.cuda.current_stream(dy.device).cuda_stream), "upsample_bilinear_ac_bwd_nhwc")
        return dx, None


class Upsample2x(nn.Upsample):
    """nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) with the native backward for channels-last
    float32 maps on the GPU."""

    def __init__(self):
        super().__init__(scale_factor=2, mode="bilinear", align_corners=True)

    def forward(self, x):
        if (USE_NATIVE_DECODER_BACKWARD and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and x.shape[1] % 4 == 0 and x.shape[2] > 1 and x.shape[3] > 1 and torch.is_grad_enabled()
                and x.requires_grad and x.is_contiguous(memory_format=torch.channels_last)):
            return _UpsampleAlignedFn.apply(x, (2 * x.shape[2], 2 * x.shape[3]))
        return super().forward(x)


def _conv3x3(cin, cout, stride=1, dilation=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=dilation, dilation=dilation, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride=stride, dilation=dilation)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes, dilation=dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        out = out + (x if self.downsample is None else self.downsample(x))
        return self.relu(out)


class ResNet(nn.Module):
    """Trunk returning (layer4 output, layer3 output), both at 1/8 resolution.  [ref resnet.py:109-202]"""

    def __init__(self, layers=(2, 2, 2, 2), num_classes=1000, output_stride=32):
        super().__init__()
        self._inplanes, self._stride, self._dilation, self._output_stride = 64, 4, 1, output_stride
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._stage(64, layers[0])
        self.layer2 = self._stage(128, layers[1], stride=2)
        self.layer3 = self._stage(256, layers[2])   # reference passes dilation=2: ignored there
        self.layer4 = self._stage(512, layers[3])   # reference passes dilation=4: ignored there
        self.avgpool = nn.AvgPool2d(7)
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _stage(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self._inplanes != planes:
            if self._stride == self._output_stride:   # never true with output_stride 32
                self._dilation *= stride
                stride = 1
            else:
                self._stride *= stride
            downsample = nn.Sequential(nn.Conv2d(self._inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes))
        seq = [BasicBlock(self._inplanes, planes, stride, downsample, dilation=self._dilation)]
        self._inplanes = planes
        seq += [BasicBlock(planes, planes, dilation=self._dilation) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer2(self.layer1(x))
        x3 = self.layer3(x)
        return self.layer4(x3), x3


class PSPModule(nn.Module):
    """Pyramid pooling at bin sizes (1,2,3,6) + 1x1 bottleneck.  [ref modules.py:10-34]"""

    def __init__(self, features, out_features=1024, sizes=(1, 2, 3, 6)):
        super().__init__()
        self.stages = nn.ModuleList([
            nn.Sequential(nn.AdaptiveAvgPool2d(output_size=(size, size)),
                          nn.Conv2d(features, features, kernel_size=1, bias=False)) for size in sizes])
        self.bottleneck = nn.Conv2d(features * (len(sizes) + 1), out_features, kernel_size=1)
        self.relu = nn.ReLU()

    def forward(self, feats):
        h, w = feats.size(2), feats.size(3)
        priors = [F.interpolate(stage(feats), size=(h, w), mode="bilinear", align_corners=False)
                  for stage in self.stages] + [feats]
        return self.relu(self.bottleneck(torch.cat(priors, 1)))


class PSPUpsample(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Sequential(Upsample2x(), nn.Conv2d(in_channels, out_channels, 3, padding=1),
                                  nn.BatchNorm2d(out_channels), PReLU())

    def forward(self, x):
        return self.conv(x)


class Modified_PSPNet(nn.Module):
    """[ref modules.py:51-81]"""

    def __init__(self, sizes=(1, 2, 3, 6), psp_size=512):
        super().__init__()
        self.feats = ResNet((2, 2, 2, 2))
        self.psp = PSPModule(psp_size, 1024, sizes)
        self.drop_1 = nn.Dropout2d(p=0.3)
        self.up_1 = PSPUpsample(1024, 256)
        self.up_2 = PSPUpsample(256, 64)
        self.up_3 = PSPUpsample(64, 64)
        self.drop_2 = nn.Dropout2d(p=0.15)
        self.final = nn.Sequential(nn.Conv2d(64, 128, kernel_size=1), nn.BatchNorm2d(128), PReLU())

    def forward(self, x, choose=None):
        """rgb (B,3,H,W) -> (B,128,H,W); with ``choose`` (B,N) flat pixel indices in eval mode -> (B,128,N), the
        features of the chosen pixels only (see ``_tail_at``)."""
        f, _ = self.feats(x)
        p = self.drop_1(self.psp(f))
        p = self.drop_2(self.up_1(p))
        p = self.drop_2(self.up_2(p))
        if choose is not None and not self.training:
            return self._tail_at(p, choose)
        return self.final(self.up_3(p))

    def _tail_at(self, p, choose):
        """Eval mode: the last decoder stage (`up_3`: 2x bilinear upsample, 3x3 conv, BatchNorm, PReLU) and `final`
        (1x1 conv, BatchNorm, PReLU) evaluated at the chosen pixels only.  IST-Net reads N of the H*W output pixels
        (ist_net.py:41-45) and every op after the upsample is local to a 3x3 window of it, so the 3x3 conv becomes a
        (B*N, 9*64) x (9*64, 64) product over gathered windows instead of a convolution over the full 192x192 map, and
        the (B,64,H,W) conv output and the (B,128,H,W) feature map (1.2 GB at B=64) are never materialised.  Same
        arithmetic per pixel as the dense path up to summation order; not valid in training mode, where BatchNorm
        takes batch statistics over all pixels."""
        upsample, conv, bn, act = self.up_3.conv[0], self.up_3.conv[1], self.up_3.conv[2], self.up_3.conv[3]
        up = upsample(p)                                             # (B, C, H, W)
        b, c, h, w = up.shape
        if not up.is_contiguous() and up.is_contiguous(memory_format=torch.channels_last):
            rows = up.permute(0, 2, 3, 1).reshape(b, h * w, c)       # view
        else:
            rows = up.reshape(b, c, h * w).transpose(1, 2)           # strided view; gather handles it
        r, q = choose // w, choose % w                               # (B, N)
        d = torch.tensor([-1, 0, 1], device=choose.device)
        rr = (r.unsqueeze(-1) + d.repeat_interleave(3)).clamp_(0, h - 1)     # (B, N, 9): window rows, kernel-row major
        qq = (q.unsqueeze(-1) + d.repeat(3)).clamp_(0, w - 1)
        inside = ((r.unsqueeze(-1) + d.repeat_interleave(3) == rr) & (q.unsqueeze(-1) + d.repeat(3) == qq))
        flat = (rr * w + qq).reshape(b, -1)                          # (B, N*9)
        win = torch.gather(rows, 1, flat.unsqueeze(-1).expand(-1, -1, c))    # (B, N*9, C)
        win = (win * inside.reshape(b, -1, 1)).reshape(b, choose.size(1), 9 * c)   # zero padding of the conv
        w2 = conv.weight.permute(0, 2, 3, 1).reshape(conv.out_channels, 9 * c)      # (Cout, kr, kc, Cin)
        y = torch.matmul(win, w2.t()) + conv.bias                    # (B, N, Cout)
        y = F.batch_norm(y.transpose(1, 2), bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
        y = F.prelu(y, act.weight)                                   # (B, Cout, N)
        conv, bn, act = self.final[0], self.final[1], self.final[2]
        y = F.conv1d(y, conv.weight.view(conv.out_channels, -1, 1), conv.bias)
        y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
        return F.prelu(y, act.weight).contiguous()


class ModifiedResnet(nn.Module):
    """`rgb_cam_extractor` of IST-Net.  [ref modules.py:232-241]"""
    gathers_choose = True     # forward(x, choose) returns the chosen pixels' features in eval mode

    def __init__(self):
        super().__init__()
        self.model = Modified_PSPNet(sizes=(1, 2, 3, 6), psp_size=512)

    def forward(self, x, choose=None):
        return self.model(x, choose)

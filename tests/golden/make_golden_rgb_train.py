"""Training-mode golden of the RGB branch (model/modules.py:10-81 ModifiedResnet = ResNet-18 trunk + PSP decoder) from the
REFERENCE's own modules: batch-statistics BatchNorm everywhere, the two Dropout2d layers switched to eval (their masks come
from the framework's random stream, which differs between devices), B = 2 images of 96 x 96.  Stored: the input, a
sub-sampled float32 output, the same from a float64 copy of the reference, and the running statistics of the decoder's last
BatchNorm after the step.  Build container only; data only.

    python tests/golden/make_golden_rgb_train.py
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    _, _, ref_model_modules, _, _ = mg.import_reference()
    import resnet as ref_resnet
    ref_resnet.model_zoo.load_url = lambda *a, **k: ref_resnet.ResNet(ref_resnet.BasicBlock, [2, 2, 2, 2]).state_dict()
    torch.manual_seed(60)
    net = ref_model_modules.ModifiedResnet()
    # non-trivial affine parameters, so that batch-statistics errors are not hidden behind gamma = 1, beta = 0
    g = torch.Generator().manual_seed(64)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    net.train()
    for m in net.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.eval()
    net64 = copy.deepcopy(net).double()
    img = torch.randn(2, 3, 96, 96, generator=g)
    out = net(img)
    with torch.no_grad():
        out64 = net64(img.double())
    last_bn = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)][-1]
    print("train-mode RGB branch: max |fp32 - fp64| / max |fp64| =",
          float((out.double() - out64).abs().max() / out64.abs().max()), "output", tuple(out.shape))
    state = {k: mg.npy(v) for k, v in net.state_dict().items() if "num_batches" not in k and "running" not in k}
    np.savez_compressed(os.path.join(HERE, "rgb_branch_train.npz"), img=mg.npy(img),
                        out_sub=mg.npy(out)[:, ::4, ::6, ::6], out_sub_f64=mg.npy(out64)[:, ::4, ::6, ::6],
                        last_running_mean=mg.npy(last_bn.running_mean), last_running_var=mg.npy(last_bn.running_var),
                        bn_seed=np.int64(64), n_state=np.int64(len(state)),
                        state_checksum=mg.state_checksum(mg.params_only(net.state_dict())))
    print(os.path.getsize(os.path.join(HERE, "rgb_branch_train.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()

import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from istnet_amd.optim import FlatAdam
from istnet_amd.pointnet2 import fused_mlp
from istnet_amd.pointnet2.pytorch_utils import BNMomentumScheduler
dev = torch.device("cuda:0")
model = bench.make_model(dev, seed=3)
ref = copy.deepcopy(model)
pts = bench.shell_cloud(4, 512, seed=5, device=dev)
sched = [0.5, 0.2, 0.05, 0.9]
opt = FlatAdam(model.parameters(), lr=0.0)
bnm = BNMomentumScheduler(model, bn_lambda=lambda it: sched[it], last_epoch=-1)
step = bench.make_graphed_step(bench.make_encoder_fwd_bwd(model, pts), opt, 1)
for it in (1, 2, 3):
    bnm.step(it); step()
torch.cuda.synchronize()
k = "SA_modules.0.mlps.0.layer0.normlayer.bn.running_var"
print("got after graph", model.state_dict()[k][:4].tolist())
fused_mlp._fusable = lambda *a, **kw: False
fused_mlp._fusable_shape = lambda *a, **kw: False
fused_mlp.USE_FUSED_FP = False
for i, m in enumerate([0.5] * 4 + sched[1:]):
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = m
    with torch.no_grad():
        ref(pts)
    torch.cuda.synchronize()
    print("ref step", i, "m", m, ref.state_dict()[k][:4].tolist(), "| model now", model.state_dict()[k][:2].tolist())

import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from istnet_amd.optim import FlatAdam
from istnet_amd.pointnet2 import fused_mlp
from istnet_amd.pointnet2.pytorch_utils import BNMomentumScheduler
dev = torch.device("cuda:0")
model = bench.make_model(dev, seed=3)
eager = copy.deepcopy(model)
pts = bench.shell_cloud(4, 512, seed=5, device=dev)
sched = [0.5, 0.2, 0.05, 0.9]
opt = FlatAdam(model.parameters(), lr=0.0)
bnm = BNMomentumScheduler(model, bn_lambda=lambda it: sched[it], last_epoch=-1)
step = bench.make_graphed_step(bench.make_encoder_fwd_bwd(model, pts), opt, 1)
def snap(m):
    return {k: v.clone() for k, v in m.state_dict().items() if k.endswith("running_var")}
hist = [snap(model)]
for it in (1, 2, 3):
    bnm.step(it); step(); torch.cuda.synchronize(); hist.append(snap(model))
# eager fused with the same schedule
opt2 = FlatAdam(eager.parameters(), lr=0.0)
bnm2 = BNMomentumScheduler(eager, bn_lambda=lambda it: sched[it], last_epoch=-1)
f = bench.make_encoder_fwd_bwd(eager, pts)
hist2 = []
for it in [0, 0, 0, 0, 1, 2, 3]:
    bnm2.step(it); opt2.zero_grad(set_to_none=True); f(); opt2.step(); torch.cuda.synchronize()
    if it or len(hist2) == 0 and False: pass
    hist2.append(snap(eager))
hist2 = hist2[3:]
keys = list(hist[0].keys())
for k in keys[:8] + keys[-2:]:
    print(k)
    for i in range(4):
        a, b = hist[i][k], hist2[i][k]
        print("   after step", i, "graph", [round(x, 6) for x in a[:3].tolist()], "eager", [round(x, 6) for x in b[:3].tolist()], "maxrel", float(((a - b).abs() / b.abs().clamp_min(1e-12)).max()))

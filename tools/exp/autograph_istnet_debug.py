import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from istnet_amd import graphed
from istnet_amd.ist_net import point_branch_side_streams
warnings.simplefilter("always")
dev = torch.device("cuda:0")
point_branch_side_streams(False)
model = bench.make_istnet(dev)
fwd_bwd = bench.make_istnet_fwd_bwd(model, bench.istnet_batch(32, 1024, 0, dev))
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
for it in range(6):
    opt.zero_grad()
    fwd_bwd()
    opt.step()
    torch.cuda.synchronize()
    print(it, graphed.STATS, flush=True)

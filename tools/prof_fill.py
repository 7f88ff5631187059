import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from istnet_amd import preprocess
dev = torch.device("cuda:0")
fr = bench.synthetic_frames(32, 1024, 0, dev)
for _ in range(12):
    preprocess.fill_missing(fr["depth"], 1000.0, 1)
torch.cuda.synchronize()

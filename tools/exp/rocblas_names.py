import torch
dev = torch.device("cuda:0")
for cin, cout, P in [(512, 512, 128), (256, 256, 256), (128, 128, 2048), (128, 256, 2048), (64, 128, 4096)]:
    x = torch.randn(32, cin, P, device=dev); w = torch.randn(cout, cin, device=dev); y = torch.empty(32, cout, P, device=dev)
    for _ in range(5): torch.matmul(w, x, out=y)
torch.cuda.synchronize()

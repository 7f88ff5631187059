"""The reference's training loop (train.py + utils/solver.py:84-100) on this package, with synthetic batches.

Same objects as the reference builds -- IST_Net with the ResNet-18/PSP RGB branch, SupervisedLoss, Adam driven by a
per-iteration CyclicLR, the BatchNorm-momentum schedule -- with the three substitutions INTEGRATION.md describes:
``FlatAdam`` for ``torch.optim.Adam``, one process per GPU + ``OverlappedFlatReducer`` for ``nn.DataParallel``, and synthetic
data for the NOCS loaders (no dataset in this repository).

    python examples/train_synthetic.py --iters 20
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_synthetic.py --iters 20
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (synthetic batch generator of SURVEY.md 8d config 3)
import istnet_amd  # noqa: E402,F401
from istnet_amd.ist_net import IST_Net, point_branch_side_streams  # noqa: E402
from istnet_amd.losses import SupervisedLoss  # noqa: E402
from istnet_amd.optim import FlatAdam, layout_hints  # noqa: E402
from istnet_amd.parallel import OverlappedFlatReducer, broadcast_parameters  # noqa: E402
from istnet_amd.pointnet2.pytorch_utils import BNMomentumScheduler  # noqa: E402
from istnet_amd.rgb_branch import ModifiedResnet  # noqa: E402
from istnet_amd import tuned_gemm  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--npoints", type=int, default=1024)
    ap.add_argument("--img", type=int, default=192)
    ap.add_argument("--freeze-world-enhancer", action="store_true")
    ap.add_argument("--eager", action="store_true", help="single GPU: do not capture the step in a HIP graph")
    args = ap.parse_args(argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    # the RGB branch fills the chip: the point branch's side streams only get in its way (restored on the way out)
    switches = point_branch_side_streams(False)
    tuned = tuned_gemm.enable()  # recorded solutions for the RGB decoder's library GEMMs (look-up only; no-op without a table)
    try:
        return _train(args, dev, world, rank)
    finally:
        switches.restore()
        if tuned:
            tuned_gemm.disable()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()


def _train(args, dev, world, rank):
    torch.manual_seed(0)
    model = IST_Net(rgb_extractor=ModifiedResnet(), freeze_world_enhancer=args.freeze_world_enhancer).to(dev).train()
    model.rgb_cam_extractor.to(memory_format=torch.channels_last)
    if args.freeze_world_enhancer:                      # train.py:116-118
        for p in model.world_enhancer.parameters():
            p.requires_grad_(False)
    if world > 1:
        broadcast_parameters(model, src=0)
    # solver.py:40-49 -- Adam (betas of config/ist_net_default.yaml), CyclicLR every iteration, BN momentum decay
    opt = FlatAdam(model.parameters(), lr=1e-5, betas=(0.5, 0.999), adjacent=layout_hints(model))
    sched = torch.optim.lr_scheduler.CyclicLR(opt, base_lr=1e-5, max_lr=1e-3, step_size_up=max(args.iters // 6, 1),
                                              mode="triangular", cycle_momentum=False)
    bnm = BNMomentumScheduler(model, bn_lambda=lambda it: max(0.5 * 0.5 ** int(it / 200000), 0.01), last_epoch=0)
    # buckets of the flat gradient buffer leave from autograd hooks while backward still runs
    reducer = OverlappedFlatReducer(opt, world) if world > 1 else None
    criterion = SupervisedLoss(1.0, 10.0, freeze_world_enhancer=args.freeze_world_enhancer)
    label_keys = ("rotation_label", "translation_label", "size_label", "qo")

    # One process, one GPU: forward + backward + Adam are captured ONCE in a HIP graph and replayed; what changes per
    # iteration reaches the replay through device memory -- the batch (copied into static tensors), the learning rate
    # (FlatAdam's device slot, ``sync_lr``) and the BatchNorm momentum (``BNMomentumScheduler.step`` writes the slots the
    # finalize kernels read).  [solver.py:88-99: lr_scheduler.step, bnm_scheduler.step, forward, backward, optimizer.step]
    graphed = world == 1 and not args.eager
    static = bench.istnet_batch(args.batch, args.npoints, seed=1000 * rank, device=dev, hw=args.img)
    holder = {}

    def fwd_bwd():
        end_points = model(static)
        end_points.update({k: static[k] for k in label_keys})
        loss = criterion(end_points)
        loss.backward()
        holder["loss"] = loss.detach()
        return loss

    replay = None
    history = []
    for it in range(args.iters):
        batch = bench.istnet_batch(args.batch, args.npoints, seed=1000 * rank + it, device=dev, hw=args.img)
        for k, v in batch.items():
            static[k].copy_(v)
        bnm.step(it)
        if graphed:
            if replay is None:      # (make_graphed_step runs four eager warm-up steps on the first batch before capturing)
                replay = bench.make_graphed_step(fwd_bwd, opt, 1)
            opt.sync_lr()           # this iteration's learning rate -> the device slot the captured Adam launch reads
            replay()
        else:
            opt.zero_grad(set_to_none=True)
            fwd_bwd()
            if reducer is not None:
                opt.step(reducer.finish(), grad_scale=1.0 / world)
            else:
                opt.step()
        sched.step()
        history.append(float(holder["loss"]))
        if rank == 0 and (it % 5 == 0 or it == args.iters - 1):
            print(f"iter {it:4d}  lr {opt.param_groups[0]['lr']:.2e}  loss {history[-1]:.4f}", flush=True)
    return history


if __name__ == "__main__":
    main()

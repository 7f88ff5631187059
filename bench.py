"""bench.py -- point clouds / s, forward + backward (+ Adam step), B=32 N=1024 per GPU.

Workload (BASELINE.json configs[1]): the IST-Net point encoder ``PointNet2MSG`` with the camera
radii of model/ist_net.py:16 -- 4 MSG set-abstraction levels + 4 feature-propagation levels
 -- train-mode BatchNorm, loss = mean(out^2), on synthetic "shell"
cloud batches (seeded).  One step = zero_grad + forward + backward (+ one RCCL all-reduce of the packed
gradients when N > 1) + fused Adam step over the flat parameter buffer (istnet_amd.optim.FlatAdam).
Inputs are resident in HBM before the timed region.

Pipelining (default; ``--no-prefetch`` turns it off): two batches alternate, and while step t runs, the
geometry stream computes the coordinate-only part (FPS, ball query, three_nn, interpolation weights) of
batch t+1 -- next-batch preprocessing, which a data loader overlaps in the same way.  Every timed step still
executes exactly one geometry pass, one forward, one backward and one optimizer update; only the first
batch's geometry is computed before the timed region (pipeline prologue).  With ``--no-prefetch`` the same
batch is used every step and its geometry is computed inside the step.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python bench.py --workload istnet      # full IST-Net training step (configs[2]/[3]): RGB branch + point branch
    python bench.py --workload infer       # eval-mode full model + post-processing, B=64 N=2048 (config 5)
    python bench.py --workload pipeline    # config 3 WITH the device-side input preparation in front (pipeline_infer: config 5)
    python bench.py --workload istnet --split-precision   # + the opt-in split-precision trunk under the key `split_precision`

A short run (steps <= 20) times five windows and reports the median.  Besides `roofline` and `cpu_baseline` the encoder line
carries `unpipelined` (the same step without the geometry prefetch) and `eager`: what an unchanged reference-style loop gets
(zero_grad / model(batch) / loss.backward() / torch.optim.Adam.step(), no whole-step graph, no prefetch).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CAM_RADII = [[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]]  # ist_net.py:16
BATCH, NPOINTS = 32, 1024
ENCODER_FWD_BWD_MFLOP_PER_CLOUD = 4394.7      # 3 x 1 464.9 MFLOP dense forward work per cloud
PMC_TRAFFIC_FILE = "r06_pmc_traffic.json"     # rocprofv3 --pmc summary the roofline object's `traffic` is read from
PREFETCH_SPLIT = ""       # make_pipelined_fwd_bwd: where the second part of the next batch's geometry is issued (see there)
PMC_SQ_FILE = "r06_pmc_sq_counters.json"      # rocprofv3 --pmc SQ counters: roofline.mfma_busy_frac (tools/pmc_sq.sh)


def shell_cloud(b, n, seed, device="cpu"):
    """Unit-normal directions x 0.1 m + N(0, 0.002) noise, centred."""
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(b, n, 3, generator=g)
    pts = d / d.norm(dim=2, keepdim=True) * 0.1 + torch.randn(b, n, 3, generator=g) * 0.002
    pts = pts - pts.mean(dim=1, keepdim=True)
    return pts.contiguous().to(device)


def cube_cloud(b, n, seed, device="cpu"):
    """U(-0.1, 0.1)^3, centred: the second distribution SURVEY.md 8d names for config 2 (balls of the fine levels mostly
    under-full, like the shell's, but no surface structure)."""
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand(b, n, 3, generator=g) * 0.2 - 0.1
    pts = pts - pts.mean(dim=1, keepdim=True)
    return pts.contiguous().to(device)


def dense_cloud(b, n, seed, device="cpu"):
    """A shell of radius 0.02 m (noise 0.0004): every ball of every level holds more points than nsample, so no row is
    padded -- compact columns at level 1 and the first-hit padding buy nothing, every grouped column is distinct work."""
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(b, n, 3, generator=g)
    pts = d / d.norm(dim=2, keepdim=True) * 0.02 + torch.randn(b, n, 3, generator=g) * 0.0004
    pts = pts - pts.mean(dim=1, keepdim=True)
    return pts.contiguous().to(device)


CLOUDS = {"shell": shell_cloud, "cube": cube_cloud, "dense": dense_cloud}


def mse_value_and_grad(out):
    """The encoder workloads' scalar loss, mean(out^2), and its gradient -- the mean-squared-error op of the reference's
    SupervisedLoss (model/ist_net.py:99) against a zero target, value and gradient from one launch."""
    from istnet_amd.losses import mse_value_and_grad as f
    return f(out)


def make_model(device, seed=0):
    from istnet_amd.modules import PointNet2MSG
    torch.manual_seed(seed)
    return PointNet2MSG([list(r) for r in CAM_RADII]).to(device).train()


def make_istnet(device, seed=0, freeze_world_enhancer=False):
    """Full IST-Net (BASELINE configs[2] / [3]): RGB branch on MIOpen + point branch on the HIP kernels.
    freeze_world_enhancer: the second training stage (train.py:102-118) -- world encoder frozen, no world pose head."""
    from istnet_amd.ist_net import IST_Net
    from istnet_amd.rgb_branch import ModifiedResnet
    torch.manual_seed(seed)
    net = IST_Net(rgb_extractor=ModifiedResnet(), freeze_world_enhancer=freeze_world_enhancer).to(device).train()
    if freeze_world_enhancer:
        for p in net.world_enhancer.parameters():
            p.requires_grad_(False)
    # MIOpen runs the 2-D convolutions 1.5x faster in NHWC (fp32 either way; tools/bench_rgb.py: 83.9 -> 56.2 ms)
    net.rgb_cam_extractor.to(memory_format=torch.channels_last)
    return net


def istnet_batch(b, n, seed, device, hw=192):
    """Synthetic RGB-D batch of SURVEY.md 8(d) config 3."""
    g = torch.Generator().manual_seed(1000 + seed)
    pts = shell_cloud(b, n, seed) + torch.tensor([0.0, 0.0, 0.8])
    rot = torch.linalg.qr(torch.randn(b, 3, 3, generator=g))[0]
    batch = {"rgb": torch.randn(b, 3, hw, hw, generator=g).contiguous(memory_format=torch.channels_last), "pts": pts,
             "choose": torch.randint(0, hw * hw, (b, n), generator=g),
             "category_label": torch.randint(0, 6, (b, 1), generator=g),
             "qo": torch.rand(b, n, 3, generator=g) - 0.5,
             "rotation_label": rot, "translation_label": pts.mean(dim=1),
             "size_label": torch.rand(b, 3, generator=g) * 0.25 + 0.05}
    return {k: v.to(device) for k, v in batch.items()}


def make_istnet_fwd_bwd(model, batch):
    from istnet_amd.losses import SupervisedLoss
    crit = SupervisedLoss(1.0, 10.0, freeze_world_enhancer=model.freeze_world_enhancer)
    labels = {k: batch[k] for k in ("rotation_label", "translation_label", "size_label", "qo")}

    def fwd_bwd():
        ep = model(batch)
        ep.update(labels)
        loss = crit(ep)
        loss.backward()
        return loss
    return fwd_bwd


def make_encoder_fwd_bwd(model, pts):
    def fwd_bwd():
        out = model(pts)
        loss, grad = mse_value_and_grad(out)
        out.backward(grad)
        return loss
    return fwd_bwd


def make_pipelined_fwd_bwd(model, batches, slots, i):
    """Step on batch i while the geometry stream prepares batch 1-i (FPS / ball query / three_nn depend on the
    coordinates only -- next-batch preprocessing, as a data loader would overlap it).  Every step still runs one
    full geometry pass; it just runs one step ahead of its consumer."""
    split = os.environ.get("ISTNET_PREFETCH_SPLIT", PREFETCH_SPLIT)

    def fwd_bwd():
        # The next batch's level-1 FPS (a 300-us chain of 32 waves: nearly no load on the chip) starts with the step; the rest
        # of its geometry (ball queries, compaction, three_nn, inverse lists: ~150 us of short bandwidth kernels) is issued where
        # the chip has room for it -- PREFETCH_SPLIT: "" = everything at once (rounds 3-5), "after_sa" = between the SA and the
        # FP levels of the forward pass (the FP phases are one dependent chain on a mostly idle chip), "after_fwd" = after it
        if split:
            finish = model.prefetch_geometry(batches[1 - i], slots[1 - i], split=True)
            if split == "after_sa":
                slots[i].after_sa = finish
            out = model(batches[i], geometry=slots[i])
            if split != "after_sa":
                finish()
        else:
            model.prefetch_geometry(batches[1 - i], slots[1 - i])
            out = model(batches[i], geometry=slots[i])
        loss, grad = mse_value_and_grad(out)      # mean of squares and its gradient 2 out / n from one launch
        out.backward(grad)
        model.join_geometry()
        return loss
    return fwd_bwd


def make_eager_step(fwd_bwds, opt, world, reducer=None):
    fwd_bwds = list(fwd_bwds) if isinstance(fwd_bwds, (list, tuple)) else [fwd_bwds]
    count = [0]

    def step():
        opt.zero_grad(set_to_none=True)
        loss = fwd_bwds[count[0] % len(fwd_bwds)]()
        count[0] += 1
        optimizer_step(opt, world, reducer)
        return loss
    return step


def optimizer_step(opt, world, reducer):
    """FlatAdam: the flat gradient buffer is summed over ranks in buckets (parallel.OverlappedFlatReducer: issued from
    autograd hooks while backward runs in an eager step, back to back after a graph replay), update with 1/world."""
    if reducer is not None:          # data parallel (or its one-rank dry run): the buckets went out during backward
        opt.step(reducer.finish(), grad_scale=1.0 / world)
    else:
        opt.step()


def make_step(model, pts, opt, world, reducer=None):
    return make_eager_step(make_encoder_fwd_bwd(model, pts), opt, world, reducer)


def make_graphed_step(fwd_bwds, opt, world, reducer=None):
    """Capture forward+backward(+Adam when single-GPU) of the step in one HIP graph and return a
    function that replays it.  Every kernel of the step (the C-ABI launches included) goes to the
    capture stream or a stream forked from it, so a replay does exactly the work of the eager step with one
    host call.  With N > 1 the gradient all-reduce and the optimizer run eagerly after the replay."""

    fwd_bwds = list(fwd_bwds) if isinstance(fwd_bwds, (list, tuple)) else [fwd_bwds]
    from istnet_amd import graphed
    auto_graph, graphed.ENABLED = graphed.ENABLED, False   # the whole step goes into ONE graph here: no per-module segments
    try:
        return _make_graphed_step(fwd_bwds, opt, world, reducer)
    finally:
        graphed.ENABLED = auto_graph


def _make_graphed_step(fwd_bwds, opt, world, reducer):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm-up off the default stream (allocator, autograd)
        for it in range(4):
            opt.zero_grad(set_to_none=True)
            fwd_bwds[it % len(fwd_bwds)]()
            optimizer_step(opt, world, reducer)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graphs = []
    # With a process group alive, ProcessGroupNCCL's watchdog thread polls its events (hipEventQuery) at any time;
    # under the default "global" capture mode that call is illegal while ANOTHER thread captures and the watchdog
    # aborts the process (measured on this stack: tools/exp/rccl_capture_modes.py).  "thread_local" restricts the
    # check to the capturing thread.
    capture_mode = "global" if reducer is None else "thread_local"
    in_graph = reducer is not None and getattr(reducer, "capture_collectives", False)
    for fwd_bwd in fwd_bwds:                           # one graph per closure (two when the batches alternate)
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        if reducer is not None:
            reducer.begin_capture()
        with torch.cuda.graph(graph, capture_error_mode=capture_mode):
            fwd_bwd()
            if reducer is None:
                opt.step()
            elif in_graph:
                # the buckets left from the backward hooks as graph nodes (forked onto the communication stream as soon as
                # their slice was final); join them and capture the Adam launch too: one host call per step
                opt.step(reducer.finish_captured(), grad_scale=1.0 / world)
            # else: the gradients stay in their (static) buffers; the all-reduce and Adam follow the replay eagerly
        # per graph: which static tensor holds each parameter's gradient after a replay (parallel.OverlappedFlatReducer)
        graphs.append((graph, reducer.end_capture() if reducer is not None else None))
    count = [0]

    def step():
        graph, token = graphs[count[0] % len(graphs)]
        count[0] += 1
        graph.replay()
        if reducer is not None and not in_graph:        # bucketed RCCL all-reduce + one Adam launch
            opt.step(reducer.finish(captured=token), grad_scale=1.0 / world)
    return step


def measure_eager(dev, workload="encoder", steps=30, warmup=8):
    """What a reference-style caller gets (utils/solver.py:88-99): ``zero_grad`` -> ``model(batch)`` -> ``loss.backward()``
    -> ``optimizer.step()`` issued from Python every step, no whole-step graph, no geometry prefetch, the loss built from
    framework ops.  Four variants: torch.optim.Adam (the reference's optimizer) and FlatAdam, each with the module-level
    HIP-graph segments of istnet_amd.graphed (the default: forward and backward of the encoder are one graph launch each
    after two warm-up calls) and launch by launch (``ISTNET_AUTO_GRAPH=0``).  ``host_ms`` is the time the Python loop
    needs to ISSUE a step (the loop's wall clock before the final synchronize): host-bound when it equals ``ms_per_step``."""
    from istnet_amd import graphed
    from istnet_amd.optim import FlatAdam, layout_hints
    out = {}
    saved = graphed.ENABLED
    try:
        for auto in (True, False):
            graphed.ENABLED = auto
            for opt_name in ("torch.optim.Adam", "FlatAdam"):
                if workload == "encoder":
                    model = make_model(dev, seed=0)
                    pts = shell_cloud(BATCH, NPOINTS, seed=0, device=dev)

                    def fwd_bwd(model=model, pts=pts):
                        loss = model(pts).square().mean()
                        loss.backward()
                else:
                    model = make_istnet(dev, seed=0)
                    fwd_bwd = make_istnet_fwd_bwd(model, istnet_batch(BATCH, NPOINTS, seed=0, device=dev))
                opt = (torch.optim.Adam(model.parameters(), lr=1e-4) if opt_name == "torch.optim.Adam"
                       else FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model)))

                def step():
                    opt.zero_grad()
                    fwd_bwd()
                    opt.step()
                for _ in range(warmup):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    step()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                ms, host = (t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3
                key = ("graph_segments" if auto else "launch_by_launch") + "/" + opt_name
                out[key] = {"ms_per_step": ms, "clouds_per_s": BATCH / ms * 1e3, "host_ms": host}
                del model, opt, fwd_bwd, step
                graphed.reset()
                torch.cuda.empty_cache()
    finally:
        graphed.ENABLED = saved
    return out


def synthetic_frames(b, n, seed, device):
    """What the reference's Dataset reads from disk for ``b`` instances (provider/dataset.py:162-200, 333-390), synthetic and
    resident in HBM: raw 480 x 640 depth in millimetres (a slanted plane, 10 % dropouts, holes, an empty band on top), the
    BGR colour image, one detection box per instance and the ``choose`` list -- the flat crop indices of n mask pixels,
    which the reference draws on the host with np.random.choice (the mask test and the draw stay with the caller)."""
    from istnet_amd import preprocess
    g = torch.Generator().manual_seed(7000 + seed)
    yy, xx = torch.meshgrid(torch.arange(480.0), torch.arange(640.0), indexing="ij")
    depth = 500.0 + 1500.0 * yy / 480 + 400.0 * xx / 640 + torch.randn(b, 480, 640, generator=g) * 4
    depth[torch.rand(b, 480, 640, generator=g) < 0.10] = 0
    for i in range(b):
        for _ in range(6):
            r, c, sz = (int(torch.randint(20, 440, (1,), generator=g)), int(torch.randint(0, 600, (1,), generator=g)),
                        int(torch.randint(4, 20, (1,), generator=g)))
            depth[i, r:r + sz, c:c + sz] = 0
    depth[:, :9] = 0
    depth = depth.clamp(0, 65535).to(torch.int32).to(torch.int16)
    image = torch.randint(0, 256, (b, 480, 640, 3), generator=g, dtype=torch.uint8)
    side = torch.randint(40, 300, (b,), generator=g)
    y1 = (torch.rand(b, generator=g) * (480 - side)).long()
    x1 = (torch.rand(b, generator=g) * (640 - side)).long()
    boxes = torch.stack([y1, x1, y1 + side, x1 + side], 1)
    win = preprocess.get_bbox(boxes)
    crop = (win[:, 1] - win[:, 0]).long()
    choose = (torch.rand(b, n, generator=g) * (crop * crop).unsqueeze(1)).long()
    return {k: v.to(device) for k, v in dict(depth=depth, image=image, boxes=boxes, choose=choose).items()}


def preprocess_batch(frames, labels, train, generator):
    """The tensor arithmetic of ``__getitem__`` for a whole batch on the device (provider/dataset.py:162-296
    for training, :333-433 for testing): depth completion, crop window, RGB crop + resize + normalise, back-projection of the
    chosen pixels + ``choose`` remap, sensor jitter, pose labels, shape augmentation.  Returns the network's input dict."""
    from istnet_amd import preprocess
    depth = preprocess.fill_missing(frames["depth"], 1000.0, 1)
    win = preprocess.get_bbox(frames["boxes"])
    rgb = preprocess.crop_resize_normalize(frames["image"], win, 192)
    pts, choose = preprocess.backproject_choose(depth, win, frames["choose"])
    out = {"rgb": rgb.contiguous(memory_format=torch.channels_last), "pts": pts, "choose": choose,
           "category_label": labels["category_label"]}
    if train:
        b = pts.shape[0]
        pts = preprocess.jitter_points(pts, generator=generator)
        sym = labels["category_label"].reshape(-1) < 3            # bottle / bowl / can stand-ins: y-symmetric classes
        rot, size, qo, _ = preprocess.instance_labels(pts, labels["translation"], labels["rotation"], labels["scale"],
                                                      labels["sizes"], sym)
        bb, rt_t, rt_r = preprocess.generate_aug_parameters(b, device=pts.device, generator=generator)
        sym_info = torch.stack([sym.long()] + [torch.zeros_like(sym, dtype=torch.long)] * 3, 1)
        pts, rot, trans, size, _, qo = preprocess.data_augment(
            preprocess.AUG_PROBS_DEFAULT, pts, rot.float(), labels["translation"], size, sym_info, bb, rt_t, rt_r,
            labels["model"], qo.float(), labels["category_label"].reshape(-1), generator=generator)
        out.update(pts=pts, qo=qo, rotation_label=rot, translation_label=trans, size_label=size)
    return out


def run_pipeline(args, dev):
    """``--workload pipeline`` / ``pipeline_infer``: the config-3 training step (or the config-5 inference batch) WITH the
    device-side input preparation in front of it, every step on fresh raw frames that are resident in HBM.  Reports the
    whole step, the preparation alone and its share."""
    train = args.workload == "pipeline"
    b, n = (BATCH, NPOINTS) if train else (64, 2048)
    from istnet_amd import postprocess
    from istnet_amd.ist_net import point_branch_side_streams
    from istnet_amd.optim import FlatAdam, layout_hints
    point_branch_side_streams(False)
    net = make_istnet(dev, seed=0)
    gen = torch.Generator().manual_seed(11)
    frames = [synthetic_frames(b, n, s, dev) for s in (0, 1)]
    g = torch.Generator().manual_seed(12)
    rot = torch.linalg.qr(torch.randn(b, 3, 3, generator=g))[0]
    labels = {k: v.to(dev) for k, v in dict(
        category_label=torch.randint(0, 6, (b, 1), generator=g), rotation=rot,
        translation=torch.tensor([0.0, 0.0, 1.2]) + torch.randn(b, 3, generator=g) * 0.05, scale=torch.rand(b, generator=g) * 0.3 + 0.1,
        sizes=torch.rand(b, 3, generator=g) * 0.5 + 0.5, model=torch.rand(b, 1024, 3, generator=g) - 0.5).items()}
    static = preprocess_batch(frames[0], labels, train, gen)          # the tensors the captured step reads
    static = {k: v.clone(memory_format=torch.preserve_format) for k, v in static.items()}
    if train:
        opt = FlatAdam(net.parameters(), lr=1e-4, adjacent=layout_hints(net))
        model_step = make_graphed_step(make_istnet_fwd_bwd(net, static), opt, 1)
    else:
        net.eval()

        def model_step():
            with torch.no_grad():
                ep = net(static)
                rts, scales = postprocess.assemble_pred_RTs(ep["pred_rotation"], ep["pred_translation"], ep["pred_size"])
                return rts.cpu(), scales.cpu()
    # The preparation is a fixed-shape sequence of ~200 small launches (most of them the batched tensor expressions of
    # data_augment): issued from Python it is host-bound (4.1 ms at B = 32, profiles/r05_preproc_stages.txt), so it is captured
    # into HIP graphs once per frame set, like the model step.  Random draws come from the device's default generator, which
    # torch advances correctly across replays.
    staged = [None, None]
    prep_graphs = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in (0, 1):
            for _ in range(2):
                preprocess_batch(frames[i], labels, train, None)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for i in (0, 1):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            staged[i] = preprocess_batch(frames[i], labels, train, None)
        prep_graphs.append(gr)
    count = [0]
    prep_stream = torch.cuda.Stream()

    def load(i):
        for k, v in staged[i].items():
            static[k].copy_(v)

    def prep():                       # preparation alone (replay + hand-over to the model's static inputs)
        i = count[0] % 2
        count[0] += 1
        prep_graphs[i].replay()
        load(i)

    def step():                       # serial: prepare this step's batch, then train on it
        prep()
        return model_step()

    def step_overlapped():            # loader-style: batch t+1 is prepared on a side stream while step t trains
        i = count[0] % 2
        count[0] += 1
        main = torch.cuda.current_stream()
        main.wait_stream(prep_stream)             # batch t is ready (prepared during step t-1)
        load(i)
        prep_stream.wait_stream(main)             # its staging buffers may be overwritten from here on
        with torch.cuda.stream(prep_stream):
            prep_graphs[1 - i].replay()
        return model_step()

    def timed(fn, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    for _ in range(args.warmup):
        step()
    n_windows = args.windows if args.windows > 0 else (5 if args.steps <= 20 else 1)
    serial = sorted(timed(step, args.steps) for _ in range(n_windows))
    dt_serial = serial[len(serial) // 2]
    dt_prep = sorted(timed(prep, args.steps) for _ in range(3))[1]
    dt_model = sorted(timed(model_step, args.steps) for _ in range(3))[1]
    with torch.cuda.stream(prep_stream):
        prep_graphs[count[0] % 2].replay()            # pipeline prologue
    for _ in range(3):
        step_overlapped()
    windows = sorted(timed(step_overlapped, args.steps) for _ in range(n_windows))
    dt_overlapped = windows[len(windows) // 2]
    # the reported step is the faster arrangement: beside the training step's convolutions the side-stream preparation
    # costs MORE than in line (its stencil kernels and the trunk compete for the same CUs); beside the inference forward less
    mode = "overlapped" if dt_overlapped < dt_serial else "serial"
    dt, windows = (dt_overlapped, windows) if mode == "overlapped" else (dt_serial, serial)
    return {"metric": ("point-clouds/sec fwd+bwd incl. input preparation, B=32 N=1024" if train else
                       "instances/sec inference incl. input preparation, B=64 N=2048"),
            "value": b / dt, "unit": "clouds/s" if train else "instances/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "windows_ms_per_step": [round(w * 1e3, 4) for w in windows],
            "config": {"workload": ("raw 480x640 depth + BGR frames resident in HBM -> fill_missing -> get_bbox -> crop / resize / "
                                    "normalise -> back-projection + choose remap"
                                    + (" -> jitter -> pose labels -> data_augment -> IST-Net training step (config 3, HIP graph)"
                                       if train else " -> IST-Net eval forward + pose post-processing + copy to the host (config 5)")),
                       "batch_per_gpu": b, "npoints": n, "global_batch": b, "parallelism": "dp1",
                       "launch": ("preparation: HIP graph replay, " + ("on a side stream one batch ahead (loader-style)" if mode == "overlapped"
                                                                           else "in line before the model step")
                                  + "; model step: " + ("HIP graph replay" if train else "eager forward"))},
            "preparation": {"ms_per_step_alone": dt_prep * 1e3, "model_step_alone_ms": dt_model * 1e3,
                            "serial_ms_per_step": dt_serial * 1e3, "share_of_serial_step": dt_prep / dt_serial,
                            "overlapped_ms_per_step": dt_overlapped * 1e3,
                            "overlapped_cost_share": (dt_overlapped - dt_model) / dt_overlapped, "reported": mode,
                            "note": "every step runs on fresh frames.  serial_*: prepare, then the model step, same stream; "
                                    "overlapped_*: batch t+1 prepared on a side stream during step t; *_alone: each part by "
                                    "itself; value / ms_per_step = the faster of the two arrangements (`reported`)"}}


def run_inference(args, dev, world, rank, dist):
    """``--workload infer`` (BASELINE configs[4] / SURVEY 8d config 5): eval-mode IST-Net, B=64 instances of
    N=2048 points + 192x192 crops per step; one step = the forward pass, the post-processing of test_func
    (solver.py:231-241) and the device->host copy of the result.  Replicas only: no collective in the step."""
    from istnet_amd import postprocess
    b, n = 64, 2048
    net = make_istnet(dev, seed=0)
    g = torch.Generator().manual_seed(5)
    for m in net.modules():          # non-trivial running statistics, as after training
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    net.eval()
    batch = istnet_batch(b, n, seed=rank, device=dev)

    def step():
        with torch.no_grad():
            ep = net(batch)
            rts, scales = postprocess.assemble_pred_RTs(ep["pred_rotation"], ep["pred_translation"], ep["pred_size"])
            return rts.cpu(), scales.cpu()

    def timed():
        for _ in range(args.warmup):
            step()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return time.perf_counter() - t0

    elapsed = timed()
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    extra = {}
    if args.split_precision and dist is None:
        # OPT-IN experiment, never the headline: the trunk's forward convolutions on the bf16 matrix pipe (three exact bf16
        # terms per fp32 operand, six products, fp32 accumulation; rgb_branch.set_split_precision)
        from istnet_amd import rgb_branch
        ref = [t.double() for t in step()]
        rgb_branch.set_split_precision(True)
        try:
            got = [t.double() for t in step()]
            dt = timed() / args.steps
        finally:
            rgb_branch.set_split_precision(False)
        extra["split_precision"] = {
            "ms_per_step": dt * 1e3, "value": b / dt, "unit": "instances/s",
            "speedup_vs_fp32": elapsed / args.steps / dt,
            "pose_max_abs_diff_vs_fp32": max(float((g_ - r_).abs().max()) for g_, r_ in zip(got, ref)),
            "scope": "ResNet trunk 3x3 / 1x1 convolutions (forward); everything else unchanged",
            "note": "opt-in (rgb_branch.set_split_precision / ISTNET_SPLIT_PRECISION=1); the value above is the exact-fp32 path"}
    return {**extra, "metric": "instances/sec inference, B=64 N=2048", "value": b * world * args.steps / elapsed,
            "unit": "instances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "IST-Net inference (eval mode; RGB branch on MIOpen with the last layer on the chosen "
                                   "pixels only, point branch on the HIP kernels, pose post-processing, result copied "
                                   "to the host)", "batch_per_gpu": b, "npoints": n, "global_batch": b * world,
                       "parallelism": f"replicas x{world}",
                       "launch": "model(inputs) under no_grad, as the reference's test loop calls it; IST_Net.forward replays its "
                                 "own per-shape HIP graph (graphed.InferenceGraph; GPU-bound at this batch either way)"}}


def run_sa_layer(args, dev):
    """``--workload sa_layer`` (BASELINE configs[0]): ONE set-abstraction layer -- furthest point
    sampling to 512 centroids, ball_query r=0.2 nsample=32, grouping, SharedMLP [3, 64, 64, 128] with train-mode
    BatchNorm, max over the ball -- forward + backward on xyz = U[0,1)^3, B=4 N=1024 (seed 0), timed on the GPU and,
    with the same modules over the CPU oracle ops, on the host cores; index tensors are compared bit-exact on the way."""
    from istnet_amd.pointnet2 import pointnet2_utils
    from istnet_amd.pointnet2.pointnet2_modules import PointnetSAModule
    from oracle import pn2_oracle
    b, n = 4, 1024
    xyz = torch.rand(b, n, 3, generator=torch.Generator().manual_seed(0))

    def make(device):
        torch.manual_seed(0)
        return PointnetSAModule(mlp=[0, 64, 64, 128], npoint=512, radius=0.2, nsample=32).to(device).train()

    def make_step(model, pts):
        def step():
            model.zero_grad(set_to_none=True)
            new_xyz, feat = model(pts)
            feat.square().mean().backward()
            return new_xyz.detach(), feat.detach()      # nothing of the autograd graph outlives the step (graph capture)
        return step

    gpu_step = make_step(make(dev), xyz.to(dev))
    for _ in range(args.warmup):
        gpu_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        new_xyz_g, feat_g = gpu_step()
    torch.cuda.synchronize()
    dt_eager = (time.perf_counter() - t0) / args.steps
    # The eager step is ~45 launches of 2-20 us kernels on B=4: what the loop above times is the host issuing them.  The
    # same step captured in a HIP graph (one host call per replay) is the device-side number and the reported value.
    dt, mode = dt_eager, "eager"
    if not args.eager:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    gpu_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                new_xyz_g, feat_g = gpu_step()
            for _ in range(args.warmup):
                graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                graph.replay()
            torch.cuda.synchronize()
            dt, mode = (time.perf_counter() - t0) / args.steps, "hipgraph"
        except Exception as exc:
            print(f"[bench] HIP graph capture of the SA layer failed ({type(exc).__name__}: {exc}); eager timing kept",
                  file=sys.stderr)
            torch.cuda.synchronize()
    # the same layer over the CPU oracle (checker + cpu_baseline leg)
    saved = pointnet2_utils._ext
    try:
        pointnet2_utils._ext = pn2_oracle
        torch.set_num_threads(min(8, os.cpu_count() or 1))
        pn2_oracle.set_threads(min(8, os.cpu_count() or 1))
        cpu_step = make_step(make("cpu"), xyz)
        for _ in range(3):
            new_xyz_c, feat_c = cpu_step()
        times = []
        for _ in range(10):
            t1 = time.perf_counter()
            cpu_step()
            times.append(time.perf_counter() - t1)
        idx_c = pn2_oracle.ball_query(new_xyz_c.contiguous(), xyz, 0.2, 32)
    finally:
        pointnet2_utils._ext = saved
    from istnet_amd.pointnet2 import _ext
    idx_g = _ext.ball_query(new_xyz_g.contiguous(), xyz.to(dev), 0.2, 32).cpu()
    parity = {"centroids_bit_exact": bool(torch.equal(new_xyz_g.cpu(), new_xyz_c)),
              "ball_query_bit_exact": bool(torch.equal(idx_g, idx_c)),
              "features_max_abs_diff": float((feat_g.detach().cpu() - feat_c.detach()).abs().max())}
    times.sort()
    cpu_dt = 0.5 * (times[4] + times[5])
    model_name, physical, logical = host_cpu()
    return {"metric": "point-clouds/sec fwd+bwd, one SA layer, B=4 N=1024", "value": b / dt, "unit": "clouds/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "single set-abstraction layer: FPS 1024->512, ball_query r=0.2 nsample=32, SharedMLP "
                                   "[3,64,64,128] train-mode BN, max-pool; fwd+bwd (BASELINE configs[0])",
                       "batch_per_gpu": b, "npoints": n, "global_batch": b, "parallelism": "dp1", "launch": mode},
            "eager_ms_per_step": dt_eager * 1e3,
            "parity": parity,
            "cpu_baseline": {"value": b / cpu_dt, "unit": "clouds/s", "cores": min(8, os.cpu_count() or 1), "kind": "port",
                             "cpu_model": model_name, "physical_cores": physical, "ms_per_step": cpu_dt * 1e3,
                             "sample": "median of 10 steps after 3 warm-up steps of the same layer (torch CPU dense "
                                       "layers + oracle/pn2_oracle.c index ops, 8 threads)"}}


def host_cpu():
    """(model string, physical cores, logical CPUs) of the host, from /proc/cpuinfo."""
    model, cores, logical = "unknown", set(), 0
    try:
        phys = core = None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                key, _, val = line.partition(":")
                key, val = key.strip(), val.strip()
                if key == "processor":
                    logical += 1
                elif key == "model name":
                    model = val
                elif key == "physical id":
                    phys = val
                elif key == "core id":
                    core = val
                elif not key and phys is not None:
                    cores.add((phys, core)); phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    logical = logical or (os.cpu_count() or 1)
    return model, (len(cores) or logical), logical


def cpu_baseline(budget_s=60.0, thread_counts=(8, 16, 32)):
    """The same step on the host cores with the CPU oracle ops (kind 'port'), to BASELINE.md section 2's protocol:
    thread-count sweep (8 / 16 / 32 threads, 1 warm-up + 2 timed steps each) keeping the fastest, then
    3 warm-up + 10 timed steps at that count, median -- cut short only if the time budget runs out (the sample string
    says what was run)."""
    from istnet_amd.pointnet2 import pointnet2_utils
    from oracle import pn2_oracle
    saved = pointnet2_utils._ext
    saved_threads = torch.get_num_threads()
    model_name, physical, logical = host_cpu()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
    t_start = time.perf_counter()
    try:
        pointnet2_utils._ext = pn2_oracle
        model = make_model("cpu")
        pts = shell_cloud(BATCH, NPOINTS, seed=0)
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)
        step = make_step(model, pts, opt, 1)

        def timed_steps(n):
            out = []
            for _ in range(n):
                t0 = time.perf_counter()
                step()
                out.append(time.perf_counter() - t0)
            return out

        sweep = {}
        # 8 / 16 / 32 threads: every box measured so far is fastest at 16 and 4-5x SLOWER with all 128 physical cores
        # (profiles/r05_cpu_baseline_all_cores.json: `--cpu-threads 128`), a multi-second step that burnt a third of
        # the driver's bench run for a number that is then discarded.
        # The process is warmed BEFORE the sweep (allocator, oneDNN primitive caches, OpenMP pools: the first two steps of
        # a process are 20-30 % slower, which the round-4 sweep charged to whichever thread count came first), and every
        # count gets its own untimed step after the pool is resized.
        counts = sorted({t for t in thread_counts if 1 <= t <= avail} or {avail})
        torch.set_num_threads(counts[0])
        pn2_oracle.set_threads(counts[0])
        step(); step()
        for nt in counts:
            torch.set_num_threads(nt)
            pn2_oracle.set_threads(nt)
            step()
            sweep[nt] = min(timed_steps(2))
            if time.perf_counter() - t_start > 0.5 * budget_s:
                break
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        pn2_oracle.set_threads(best)
        for _ in range(3):
            step()
        times = []
        while len(times) < 10 and (len(times) < 3 or time.perf_counter() - t_start < budget_s):
            times += timed_steps(1)
        times.sort()
        median = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
        # ONE statistic for the sweep and the reported value: the fastest warmed step (the box's other tenants and the
        # allocator move a 1.1-s CPU step by 20 %; the sweep already keeps the minimum of its two steps).  The chosen count's
        # sweep entry is folded in, so `value` and `thread_sweep_ms_per_step[cores]` agree by construction.
        dt = min(times[0], sweep[best])
        sweep[best] = dt
    finally:
        pointnet2_utils._ext = saved
        torch.set_num_threads(saved_threads)
    return {"value": BATCH / dt, "unit": "clouds/s", "cores": best, "kind": "port",
            "cpu_model": model_name, "physical_cores": physical, "logical_cpus": logical, "cpus_available": avail,
            "thread_sweep_ms_per_step": {str(k): round(v * 1e3, 1) for k, v in sweep.items()},
            "sample": f"fastest of {len(times)} timed steps (+ the sweep's two) after 3 warm-up steps at {best} threads "
                      f"(fastest of the sweep {sorted(sweep)}; more threads are slower on every box measured: "
                      f"profiles/r05_cpu_baseline_all_cores.json), the same B={BATCH} N={NPOINTS} encoder fwd+bwd+Adam step "
                      f"(torch CPU dense layers + oracle/pn2_oracle.c index ops, OpenMP)",
            "ms_per_step": dt * 1e3, "median_ms_per_step": median * 1e3}


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: re-run this command as N ranks (one process per GPU) under
    ``torch.distributed.run`` on 127.0.0.1 with a free port; rank 0 of the children prints the one JSON line, which
    passes through this process's stdout.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--windows", type=int, default=0,
                    help="timed windows of --steps steps each, the median is reported (default: 5 when steps <= 20, else 1)")
    ap.add_argument("--no-eager-leg", action="store_true", help="skip the reference-style eager measurement (the `eager` object)")
    ap.add_argument("--split-precision", action="store_true",
                    help="istnet / infer: ALSO measure the opt-in split-precision trunk (bf16 x 3 on the bf16 matrix pipe) and report it "
                         "under the extra key `split_precision`; the headline value stays on the exact-fp32 path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", default="", help="cpu_baseline: comma-separated thread counts to sweep (default 8,16,32)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not capture the step in a HIP graph")
    ap.add_argument("--no-unpipelined", action="store_true", help="skip the extra unpipelined measurement")
    ap.add_argument("--overlap-allreduce", action="store_true",
                    help="N > 1: run the step eagerly so that each ~25 MB bucket of the flat gradient is all-reduced "
                         "from an autograd hook while backward still runs (a replayed graph issues the buckets after "
                         "the replay); meant for --workload istnet (107 MB of gradients)")
    ap.add_argument("--capture-allreduce", action="store_true",
                    help="N > 1 (or --force-dist): capture the bucketed all-reduce and the Adam launch INSIDE the HIP graph, "
                         "each bucket forked onto the communication stream when its slice is final (overlaps the tail of "
                         "backward without host launches); falls back to replay + eager buckets if the capture fails")
    ap.add_argument("--no-overlap-allreduce", action="store_true",
                    help="(the default since round 3; kept for old command lines) captured step, buckets after the replay")
    ap.add_argument("--cloud", default="shell", choices=sorted(CLOUDS),
                    help="encoder workload: input distribution (shell = the headline; cube, dense: SURVEY 8d's other cases)")
    ap.add_argument("--no-other-clouds", action="store_true",
                    help="skip the extra cube / dense measurements of the default encoder line (`other_distributions`)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="encoder workload: one batch, geometry inside the step (no next-batch geometry prefetch)")
    ap.add_argument("--workload", default="encoder", choices=["encoder", "istnet", "infer", "sa_layer", "pipeline", "pipeline_infer"],
                    help="encoder = BASELINE configs[1] (the headline metric); istnet = full model, configs[2]/[3]; "
                         "infer = eval-mode full model + post-processing, B=64 N=2048 (config 5); sa_layer = configs[0]: "
                         "one set-abstraction layer (ball_query r=0.2, nsample=32) on B=4 N=1024, GPU beside the CPU path")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--freeze-world-enhancer", action="store_true",
                    help="istnet workload: second training stage (world encoder frozen, 23.6 M of 26.8 M parameters trained)")
    ap.add_argument("--force-dist", action="store_true",
                    help="dry run of the N>1 code path (process group, all-reduce, eager Adam after the replay) with "
                         "one rank: exercises RCCL + HIP-graph capture on a 1-GPU box")
    ap.add_argument("--same-device", action="store_true",
                    help="dry run of the N>1 control flow on a 1-GPU box: every rank uses cuda:0 (gloo only)")
    ap.add_argument("--tune-gemms", action="store_true",
                    help="istnet / infer: search the library-GEMM solutions of the RGB decoder with TunableOp and record "
                         "them in gpurun_out/tunableop_gfx950.csv (minutes; copy the table to ist-net_amd/tuning/)")
    ap.add_argument("--no-tuned-gemms", action="store_true",
                    help="istnet / infer: the library's heuristic GEMM solutions instead of the recorded table")
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="NOT a measurement: run the launch / rendezvous / barrier / all-reduce / rank-0-JSON control "
                         "flow of --gpus N on host cores (gloo, B=2, the cpu_baseline port over the oracle ops) so the "
                         "driver's command form can be tested on a box without GPUs")
    args = ap.parse_args()
    if args.cpu_dry_run:
        args.backend, args.eager, args.no_prefetch = "gloo", True, True
        args.no_roofline = args.no_cpu_baseline = args.no_unpipelined = True

    if os.environ.get("ISTNET_PW_TUNE"):   # experiments: "key:value,..." for istnet_pw_set_tuning
        from istnet_amd import _native
        for kv in os.environ["ISTNET_PW_TUNE"].split(","):
            k, v = kv.split(":")
            assert _native.lib().istnet_pw_set_tuning(int(k), int(v)) == 0
    if os.environ.get("ISTNET_PN2_TUNE"):   # experiments: "key:value,..." for istnet_pn2_set_tuning
        from istnet_amd import _native
        for kv in os.environ["ISTNET_PN2_TUNE"].split(","):
            k, v = kv.split(":")
            assert _native.lib().istnet_pn2_set_tuning(int(k), int(v)) == 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if args.same_device:
        local_rank = 0
    batch_size = BATCH
    if args.cpu_dry_run:
        from istnet_amd.pointnet2 import pointnet2_utils
        from oracle import pn2_oracle
        pointnet2_utils._ext = pn2_oracle      # checker ops stand in for the HIP library in the dry run only
        torch.set_num_threads(max(1, min(4, (os.cpu_count() or 2) // max(args.gpus, 1))))
        dev, batch_size = torch.device("cpu"), 2
        if args.workload not in ("encoder", "istnet"):
            raise SystemExit("--cpu-dry-run covers the encoder and istnet workloads")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    sync = (lambda: None) if args.cpu_dry_run else torch.cuda.synchronize

    import torch.distributed as dist
    from istnet_amd.optim import FlatAdam, layout_hints
    grad_sync = None
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:      # --force-dist without a launcher
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner to the C-level stdout when its first communicator comes up; the driver reads
        # ONE JSON line from rank 0's stdout, so fd 1 points at stderr until the communicator exists
        import ctypes
        libc = ctypes.CDLL(None)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
            else:
                dist.init_process_group(args.backend)
            warm = torch.ones(1, device=dev)
            dist.all_reduce(warm)            # forces communicator creation
            sync()
            libc.fflush(None)
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    tuned_table = None
    if args.workload in ("istnet", "infer", "pipeline", "pipeline_infer") and not args.cpu_dry_run and not args.no_tuned_gemms:
        from istnet_amd import tuned_gemm
        if args.tune_gemms:
            os.makedirs("gpurun_out", exist_ok=True)
            out = os.path.join("gpurun_out", "tunableop_gfx950.csv")
            if os.path.isfile(tuned_gemm.DEFAULT_TABLE) and not os.path.isfile(out):
                import shutil
                shutil.copyfile(tuned_gemm.DEFAULT_TABLE, out)      # new shapes are added to the recorded ones
            tuned_table = tuned_gemm.enable(out, tune=True)
        else:
            tuned_table = tuned_gemm.enable()
    if args.workload == "sa_layer":
        if dist_on:
            raise SystemExit("--workload sa_layer is a single-GPU parity / timing case (BASELINE configs[0])")
        print(json.dumps(run_sa_layer(args, dev)), flush=True)
        return
    if args.workload in ("pipeline", "pipeline_infer"):
        if dist_on:
            raise SystemExit("--workload pipeline is a single-GPU measurement of the input preparation's share")
        print(json.dumps(run_pipeline(args, dev)), flush=True)
        return
    if args.workload == "infer":
        result = run_inference(args, dev, world, rank, dist if dist_on else None)
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(result), flush=True)
        return
    npoints = NPOINTS
    if args.workload == "istnet":
        if not args.cpu_dry_run:
            from istnet_amd.ist_net import point_branch_side_streams
            point_branch_side_streams(False)      # beside the RGB branch the point branch is better off on one stream
        model = make_istnet(dev, seed=0, freeze_world_enhancer=args.freeze_world_enhancer)
        if args.cpu_dry_run:         # toy size: the launch / exchange control flow is what the dry run exercises
            npoints = 256
            batch = istnet_batch(batch_size, npoints, seed=rank, device=dev, hw=64)
            batch["rgb"] = batch["rgb"].contiguous()
            model = model.to(memory_format=torch.contiguous_format)
        else:
            batch = istnet_batch(BATCH, NPOINTS, seed=rank, device=dev)
        opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
        if dist_on:
            from istnet_amd.parallel import OverlappedFlatReducer
            grad_sync = OverlappedFlatReducer(opt, world, always=args.force_dist,
                                              capture_collectives=args.capture_allreduce)
        fwd_bwd = make_istnet_fwd_bwd(model, batch)
        args.no_cpu_baseline = True
    else:
        model = make_model(dev, seed=0)  # identical weights on every rank (same seed)
        make_cloud = CLOUDS[args.cloud]
        pts = make_cloud(batch_size, NPOINTS, seed=rank, device=dev)
        opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=layout_hints(model))
        if dist_on:
            from istnet_amd.parallel import OverlappedFlatReducer
            grad_sync = OverlappedFlatReducer(opt, world, always=args.force_dist,
                                              capture_collectives=args.capture_allreduce)
        if args.no_prefetch:
            fwd_bwd = make_encoder_fwd_bwd(model, pts)
        else:
            # two batches alternate; while a step runs, the geometry stream prepares the other batch
            from istnet_amd.modules import GeometrySlot
            batches = [pts, make_cloud(BATCH, NPOINTS, seed=1000 + rank, device=dev)]
            slots = [model.prefetch_geometry(bt, GeometrySlot()) for bt in batches]   # pipeline prologue (untimed)
            fwd_bwd = [make_pipelined_fwd_bwd(model, batches, slots, i) for i in (0, 1)]
    eager_step = make_eager_step(fwd_bwd, opt, world, grad_sync)
    step, mode = eager_step, "eager"
    # istnet with N > 1: 107 MB of gradients in ~4 buckets.  Issued from autograd hooks during an EAGER backward they hide
    # under it, but since round 3 the eager full-model step is host-bound (37.9 ms with the hooks against 33.2 ms for the
    # graph replay followed by the buckets back to back, one-rank RCCL group on a 1-GPU box: profiles/r03_bench_istnet_*),
    # and the exposed exchange is ~1 ms at N = 8 (107 MB ring all-reduce over xGMI).  So the replay + back-to-back buckets is
    # the default again; --overlap-allreduce selects the hook-overlapped eager step.
    if args.overlap_allreduce and dist_on:
        args.eager = True       # hooks issue the collectives during backward: not inside a capture
    if not args.eager and args.capture_allreduce and grad_sync is not None:
        try:
            step, mode = make_graphed_step(fwd_bwd, opt, world, grad_sync), "hipgraph"
        except Exception as exc:  # the collectives could not be captured: replay + eager buckets instead
            print(f"[bench] capturing the all-reduce failed ({type(exc).__name__}: {exc}); buckets after the replay",
                  file=sys.stderr)
            torch.cuda.synchronize()
            grad_sync.capture_collectives = False
            args.capture_allreduce = False
    if not args.eager and mode != "hipgraph":
        try:
            step, mode = make_graphed_step(fwd_bwd, opt, world, grad_sync), "hipgraph"
        except Exception as exc:  # capture unsupported in this configuration: run the same step eagerly
            print(f"[bench] HIP graph capture failed ({type(exc).__name__}: {exc}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            # eager launches are host-bound (~13 us each): the extra streams only add host work there, so the
            # fallback runs the single-stream, unpipelined step (measured 5.7 vs 7.2 ms/step)
            from istnet_amd.pointnet2 import fused_mlp
            fused_mlp.USE_SCALE_STREAMS = fused_mlp.USE_DEFERRED_WGRAD = False
            if args.workload == "encoder":
                eager_step = make_eager_step(make_encoder_fwd_bwd(model, pts), opt, world, grad_sync)
                args.no_prefetch = True
            step, mode = eager_step, "eager"

    for _ in range(args.warmup):
        step()
    # A window = exactly ``steps`` steps between barrier + synchronize on both sides, max over ranks.  A short window
    # (20 steps = 50 ms for the encoder) moves by +-1 % with the box's clocks, so short runs take several windows back
    # to back and report the MEDIAN window; every window is in the JSON line.
    n_windows = args.windows if args.windows > 0 else (5 if args.steps <= 20 and not args.cpu_dry_run else 1)
    windows = []
    for _ in range(n_windows):
        if dist_on:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        if dist_on:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        windows.append(elapsed)
    elapsed = sorted(windows)[len(windows) // 2]

    result = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        result = {
            "metric": "point-clouds/sec fwd+bwd, B=32 N=1024",
            "value": batch_size * world * args.steps / elapsed,
            "unit": "clouds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "windows_ms_per_step": [round(w / args.steps * 1e3, 4) for w in windows],
            "window_statistic": "median of %d windows of %d steps" % (len(windows), args.steps),
            "data": "synthetic" if not args.cpu_dry_run else "synthetic (CPU dry run of the launch path: NOT a measurement)",
            **({"debug_ptrs": [hex(opt.flat_grad.data_ptr()), hex(pts.data_ptr()),
                               hex(torch.empty(1 << 20, device=dev).data_ptr())]}
               if os.environ.get("ISTNET_BENCH_DEBUG_PTRS") and args.workload == "encoder" else {}),
            "config": {"workload": ("PointNet2MSG encoder (4 SA-MSG + 4 FP, cam radii) fwd+bwd+Adam, "
                                    "train-mode BN, %s clouds" % args.cloud) if args.workload == "encoder" else
                                   ("IST-Net full model (ResNet-18/PSP RGB branch on MIOpen + point branch, "
                                    "cam + world encoders, IST head, 3 pose heads) fwd+bwd+Adam, SupervisedLoss"
                                    + (", world enhancer frozen" if args.freeze_world_enhancer else "")),
                       "batch_per_gpu": batch_size, "npoints": npoints, "global_batch": batch_size * world,
                       "parallelism": f"dp{world}" + (" (one-rank dry run of the RCCL path)" if args.force_dist and world == 1 else ""),
                       "launch": mode,
                       "library_gemms": (None if args.workload == "encoder" else
                                         ("TunableOp look-up, " + os.path.basename(tuned_table)) if tuned_table
                                         else "library heuristic"),
                       "gradient_exchange": (None if not dist_on else
                                             {"bytes_per_step": int(opt.flat_grad.numel() * opt.flat_grad.element_size()),
                                              "buckets": len(grad_sync.buckets),
                                              "bucket_bytes": [int((hi - lo) * opt.flat_grad.element_size())
                                                               for lo, hi, _ in grad_sync.buckets],
                                              "collective": "sum all-reduce of FlatAdam.flat_grad slices (RCCL), 1/N folded into Adam",
                                              "issued": ("from autograd hooks during backward (overlapped)" if mode == "eager"
                                                         else "inside the HIP graph, each bucket forked when its slice is final; "
                                                              "Adam captured too" if args.capture_allreduce
                                                         else "back to back after the graph replay")}),
                       "batches": ("1 (same batch every step)" if (args.workload != "encoder" or args.no_prefetch)
                                   else "2 alternating, next batch's FPS/ball-query/three_nn prefetched on the "
                                        "geometry stream during the current step")},
        }
        if not dist_on and args.workload == "encoder" and not args.no_prefetch and mode == "hipgraph" \
                and not args.no_unpipelined:
            # the same step WITHOUT the next-batch geometry prefetch (one batch, geometry inside the step), for
            # reference: what the pipelining is worth, and the number to compare with an unpipelined loop
            plain = make_graphed_step(make_encoder_fwd_bwd(model, pts), opt, world, grad_sync)
            for _ in range(min(args.warmup, 5)):
                plain()
            dts = []
            for _ in range(n_windows):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    plain()
                torch.cuda.synchronize()
                dts.append((time.perf_counter() - t1) / args.steps)
            dt = sorted(dts)[len(dts) // 2]
            result["unpipelined"] = {"value": BATCH / dt, "unit": "clouds/s", "ms_per_step": dt * 1e3,
                                     "windows_ms_per_step": [round(d * 1e3, 4) for d in dts],
                                     "note": "one batch, FPS / ball query / three_nn inside the step (--no-prefetch)"}
        if (not dist_on and args.workload == "encoder" and mode == "hipgraph" and not args.no_other_clouds
                and not args.cpu_dry_run and args.cloud == "shell"):
            # SURVEY 8d names two input distributions for config 2 and the step is data-dependent (compact columns at SA
            # level 1, tie-free prefix FPS): the same captured step on the other inputs, same timing protocol
            from istnet_amd.modules import GeometrySlot
            other = {}
            for kind in ("cube", "dense"):
                try:
                    # a model decides ONCE, from its first passes, whether level 1 keeps its compact columns
                    # (PointNet2MSG._compact_probe); this leg stands for a user whose data looks like `kind` from the start
                    model.__dict__.pop("_compact_off", None)
                    model.__dict__.pop("_compact_fill", None)
                    bts = [CLOUDS[kind](BATCH, NPOINTS, seed=s, device=dev) for s in (rank, 1000 + rank)]
                    if args.no_prefetch:
                        fb = make_encoder_fwd_bwd(model, bts[0])
                    else:
                        sl = [model.prefetch_geometry(bt, GeometrySlot()) for bt in bts]
                        fb = [make_pipelined_fwd_bwd(model, bts, sl, i) for i in (0, 1)]
                    st_k = make_graphed_step(fb, opt, world, grad_sync)
                    for _ in range(min(args.warmup, 5)):
                        st_k()
                    dts = []
                    for _ in range(n_windows):
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        for _ in range(args.steps):
                            st_k()
                        torch.cuda.synchronize()
                        dts.append((time.perf_counter() - t1) / args.steps)
                    dt = sorted(dts)[len(dts) // 2]
                    other[kind] = {"ms_per_step": dt * 1e3, "value": BATCH / dt, "unit": "clouds/s",
                                   "windows_ms_per_step": [round(d * 1e3, 4) for d in dts],
                                   "compact_columns_level1": not model.__dict__.get("_compact_off", {}).get(0, False),
                                   "compact_fill_seen": [round(f, 3) for f in model.__dict__.get("_compact_fill", {}).get(0, [])]}
                    if not args.no_prefetch and not args.no_unpipelined:
                        pl = make_graphed_step(make_encoder_fwd_bwd(model, bts[0]), opt, world, grad_sync)
                        for _ in range(min(args.warmup, 5)):
                            pl()
                        dts = []
                        for _ in range(n_windows):
                            torch.cuda.synchronize()
                            t1 = time.perf_counter()
                            for _ in range(args.steps):
                                pl()
                            torch.cuda.synchronize()
                            dts.append((time.perf_counter() - t1) / args.steps)
                        other[kind]["unpipelined_ms_per_step"] = sorted(dts)[len(dts) // 2] * 1e3
                        del pl
                    del st_k, fb
                except Exception as exc:      # an extra leg must never cost the headline line
                    other[kind] = {"error": f"{type(exc).__name__}: {exc}"}
                    torch.cuda.synchronize()
            model.__dict__.pop("_compact_off", None)        # back to the headline's data: the legs below probe again
            model.__dict__.pop("_compact_fill", None)
            other["note"] = ("same model, same captured step, other inputs: cube = U(-0.1, 0.1)^3 (SURVEY 8d config 2's second "
                             "distribution); dense = shell of radius 0.02 (every ball over-full: no padded rows, nothing for the "
                             "compact columns or the first-hit padding to save: the model's first passes see that and evaluate level 1 on the padded "
                             "columns, fused_mlp.COMPACT_POLICY = auto)")
            result["other_distributions"] = other
        if not dist_on and not args.no_eager_leg and not args.cpu_dry_run and mode == "hipgraph":
            # the step an unchanged reference-style loop gets (no whole-step graph, no prefetch, torch.optim.Adam)
            try:
                result["eager"] = measure_eager(dev, args.workload)
            except Exception as exc:      # an extra leg must never cost the headline line
                result["eager"] = {"error": f"{type(exc).__name__}: {exc}"}
                torch.cuda.synchronize()
        if isinstance(result.get("eager"), dict) and "error" not in result["eager"]:
            ref = result.get("unpipelined", result)["ms_per_step"]
            result["eager"]["note"] = (
                "reference-style loop (utils/solver.py:88-99: zero_grad, model(batch), loss.backward(), optimizer.step()) "
                "issued from Python each step; graph_segments = istnet_amd.graphed (default: the encoder's forward and "
                "backward replay lazily captured HIP graphs), launch_by_launch = ISTNET_AUTO_GRAPH=0; ratio_to_graph_replay "
                "is against the whole-step graph of the same un-prefetched step")
            result["eager"]["ratio_to_graph_replay"] = result["eager"]["graph_segments/torch.optim.Adam"]["ms_per_step"] / ref
        if not dist_on and not args.no_roofline:
            from istnet_amd import roofline
            capture = None
            if mode == "hipgraph":
                n_graphs = len(fwd_bwd) if isinstance(fwd_bwd, (list, tuple)) else 1

                def capture():          # the same captured step(s) with timing events around every GEMM launch
                    timed_step = make_graphed_step(fwd_bwd, opt, world, grad_sync)
                    return lambda: [timed_step() for _ in range(n_graphs)]
            result["roofline"] = roofline.measure(
                eager_step, traffic_file=os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE),
                sq_file=os.path.join(ROOT, "profiles", PMC_SQ_FILE), capture=capture,
                steps_per_replay=(len(fwd_bwd) if isinstance(fwd_bwd, (list, tuple)) else 1))
            if result["roofline"] is not None and args.workload == "encoder":
                # whole-step rate: the encoder's dense work (4 394.7 MFLOP per cloud, forward + backward) over
                # the step time, against the fp32 MFMA peak -- includes every non-GEMM kernel and every gap
                flop = ENCODER_FWD_BWD_MFLOP_PER_CLOUD * 1e6 * batch_size
                result["roofline"]["step_flops_tflops"] = flop / (ms * 1e-3) / 1e12
                result["roofline"]["step_flops_frac"] = flop / (ms * 1e-3) / 1e12 / roofline.PEAK_MFMA_F32_TFLOPS
                result["roofline"]["nominal_gflop_per_step"] = flop / 1e9
                ex = result["roofline"].get("executed_gemm_gflop_per_step")
                if ex:
                    # the matrix pipe's own occupation over the step: executed (not nominal) GEMM flops / step time / peak
                    result["roofline"]["executed_flops_frac"] = ex * 1e9 / (ms * 1e-3) / 1e12 / roofline.PEAK_MFMA_F32_TFLOPS
        if args.split_precision and not dist_on and args.workload == "istnet" and mode == "hipgraph":
            # OPT-IN experiment, never the headline: the same step with the trunk's forward / backward-data products on the
            # bf16 matrix pipe (three exact bf16 terms per fp32 operand, six products, fp32 accumulation; rgb_branch.set_split_precision)
            from istnet_amd import rgb_branch

            def rgb_features():
                torch.manual_seed(123)                                   # the decoder's Dropout2d masks
                with torch.enable_grad():
                    return model.rgb_cam_extractor(batch["rgb"], batch["choose"]).detach().double()
            ref = rgb_features()
            rgb_branch.set_split_precision(True)
            try:
                got = rgb_features()
                split_step = make_graphed_step(fwd_bwd, opt, world, grad_sync)
                for _ in range(args.warmup):
                    split_step()
                dts = []
                for _ in range(n_windows):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        split_step()
                    torch.cuda.synchronize()
                    dts.append((time.perf_counter() - t1) / args.steps)
            finally:
                rgb_branch.set_split_precision(False)
            dt = sorted(dts)[len(dts) // 2]
            result["split_precision"] = {
                "ms_per_step": dt * 1e3, "value": batch_size / dt, "unit": "clouds/s", "speedup_vs_fp32_mfma": ms / (dt * 1e3),
                "rgb_features_max_rel_diff_vs_fp32_mfma": float((got - ref).abs().max() / ref.abs().max()),
                "scope": "ResNet trunk 3x3 / 1x1 convolutions: forward and backward-data (stride 1); everything else unchanged",
                "kernel_level_errors_vs_float64": "profiles/r05_split_precision_conv.txt (0.8-1.0x the fp32 MFMA kernel's)",
                "note": "opt-in (rgb_branch.set_split_precision / ISTNET_SPLIT_PRECISION=1); the headline value above is the exact-fp32 MFMA path"}
        if not dist_on and not args.no_cpu_baseline:
            result["cpu_baseline"] = (cpu_baseline() if not args.cpu_threads else
                                      cpu_baseline(180.0, tuple(int(t) for t in args.cpu_threads.split(","))))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)       # anything native code left in the C stdout buffer goes out first
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

"""Adam over one flat parameter buffer.

The reference trains with ``torch.optim.Adam(model.parameters(), lr)`` (train.py:101-107 through
solver.py).  The encoder has 96 small parameter tensors (the full model 400+): a multi-tensor optimizer
spends its time walking tensor lists, and the data-parallel exchange has to pack and unpack them.  Here all
parameters are re-pointed into ONE contiguous fp32 buffer at construction, the gradients are packed by one
``torch.cat`` (or arrive already packed and averaged from ``parallel.GradAllReducer``), and the update is one
launch of a flat elementwise kernel (csrc/optim.hip) -- the arithmetic of ``torch.optim.Adam`` (no amsgrad, L2
weight decay), without the tensor-list walk; PyTorch's multi-tensor fused Adam on a single 1.3 M-element tensor
runs 20 workgroups (46 us), the flat kernel 1 280 (a few us).
State lives on the device, so a step can be captured in a HIP graph.
"""
import torch


def layout_hints(model):
    """Parameter groups a model wants adjacent in a flat buffer (``FlatAdam(..., adjacent=layout_hints(model))``): every
    module may offer ``flat_layout_hints()`` returning lists of parameters."""
    hints = []
    for m in model.modules():
        f = getattr(m, "flat_layout_hints", None)
        if f is not None:
            hints += [list(g) for g in f()]
    return hints


class FlatAdam(torch.optim.Optimizer):
    """A ``torch.optim.Optimizer`` (so ``torch.optim.lr_scheduler`` classes accept it -- the reference drives Adam
    with ``CyclicLR``, utils/solver.py:41-47) with ONE parameter group: ``param_groups[0]['lr']`` is what schedulers
    write and what ``step()`` reads."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, adjacent=None):
        """``adjacent``: lists of parameters to lay out back to back in the flat buffer, in the order given (each list at
        the place of its first member); ``layout_hints(model)`` collects them.  Kernels that want several weight matrices
        as one (the layer-0 weights of the scales of a set-abstraction level: the level-wide feature-gradient product)
        then take a view of the buffer instead of packing a copy every step.  The optimizer state (``state_dict``) stays
        in parameter order whatever the layout."""
        params = list(params)
        if params and isinstance(params[0], dict):
            raise ValueError("FlatAdam: one parameter group only (pass the parameters, not a list of groups)")
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("FlatAdam: all parameters must share one device and dtype")
        index = {id(p): i for i, p in enumerate(self.params)}
        group_of = {}
        for grp in (adjacent or []):
            members = [index[id(p)] for p in grp if id(p) in index]
            if len(members) > 1 and not any(i in group_of for i in members):
                for i in members:
                    group_of[i] = members
        self.layout, placed = [], set()             # parameter indices in memory order
        for i in range(len(self.params)):
            for j in group_of.get(i, [i]):
                if j not in placed:
                    placed.add(j)
                    self.layout.append(j)
        # A parameter keeps its physical layout inside the buffer: a channels-last convolution weight (the RGB branch runs
        # channels-last on MIOpen) stays channels-last -- re-pointed as a plain contiguous view it would be converted back by
        # a copy kernel in every convolution call, forward and backward (52 launches per full-model step).
        self._strides = [self._physical_strides(p) for p in self.params]
        self.flat = torch.cat([self._physical_flat(self.params[i].detach(), self._strides[i]) for i in self.layout])
        self.offsets, off = [0] * len(self.params), 0
        for i in self.layout:                       # parameters become views of the flat buffer
            p = self.params[i]
            self.offsets[i] = off
            p.data = self._view(self.flat, i)
            off += p.numel()
        # Gradients are produced in place: the fused backward kernels write dW / dgamma / dbeta of a parameter
        # straight into its slot of this buffer (fused_mlp._grad_dest), so step() and the data-parallel
        # all-reduce use it as is -- no pack.  Parameters whose gradient arrives some other way are packed.
        self.flat_grad = torch.zeros_like(self.flat)
        for p, o in zip(self.params, self.offsets):
            p._istnet_grad_slot = self.flat_grad[o:o + p.numel()]
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = torch.zeros((), dtype=torch.float32, device=dev)
        self._ticket = torch.zeros((), dtype=torch.int32, device=dev)       # istnet_adam_step_counting: 0 between launches
        # the learning rate lives on the device: a per-iteration schedule writes this scalar (``opt.lr = value``) and
        # a step captured in a HIP graph picks the new value up without re-capture
        self._lr_dev = torch.full((), float(lr), dtype=torch.float32, device=dev)
        self._lr = float(lr)
        self.betas, self.eps, self.weight_decay = tuple(betas), float(eps), float(weight_decay)
        super().__init__(self.params, dict(lr=float(lr), betas=tuple(betas), eps=float(eps),
                                           weight_decay=float(weight_decay)))

    @staticmethod
    def _physical_strides(p):
        if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
            return tuple(p.stride())
        return None                                  # row-major contiguous in the buffer

    @staticmethod
    def _physical_flat(t, strides):
        """The elements of ``t`` in the order its slot of the flat buffer stores them."""
        if strides is None:
            return t.reshape(-1)
        return t.permute(0, 2, 3, 1).reshape(-1)     # channels-last: NHWC order is the memory order (a view, no copy)

    def _view(self, buf, i):
        """Parameter ``i``'s slot of a flat buffer, shaped and strided like the parameter."""
        p, o = self.params[i], self.offsets[i]
        if self._strides[i] is None:
            return buf[o:o + p.numel()].view(p.shape)
        return buf.as_strided(p.shape, self._strides[i], o)

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        self.param_groups[0]["lr"] = self._lr
        self._lr_dev.fill_(self._lr)

    def sync_lr(self):
        """Copy ``param_groups[0]['lr']`` (what an lr scheduler just wrote) to the device scalar the update kernel
        reads.  ``step()`` does this itself; call it before replaying a HIP graph that captured ``step()``."""
        if float(self.param_groups[0]["lr"]) != self._lr:
            self.lr = float(self.param_groups[0]["lr"])

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def grads_in_place(self):
        """True when every parameter's .grad is its slot of ``flat_grad`` (written there by the backward kernels)."""
        return all(p.grad is not None and p.grad.data_ptr() == p._istnet_grad_slot.data_ptr()
                   and p.grad.is_contiguous() for p in self.params)

    def pack_grads(self):
        """The flat gradient buffer in the buffer's layout, complete: gradients the backward kernels wrote in place are
        already there; the others (the RGB branch's convolution gradients come from MIOpen) are copied into their slots by
        one multi-tensor copy, and the slot of a parameter without a gradient is zeroed.  (Until round 3 this concatenated
        every gradient of the model whenever one of them was not in place: 2 x 107 MB at the end of a full-model step.)"""
        src, dst, zero = [], [], []
        for i, p in enumerate(self.params):
            slot = p._istnet_grad_slot
            if p.grad is None:
                zero.append(slot)
            elif p.grad.data_ptr() != slot.data_ptr() or not p.grad.is_contiguous():
                src.append(p.grad)
                dst.append(self._view(self.flat_grad, i))
        if dst:
            torch._foreach_copy_(dst, src)
        if zero:
            torch._foreach_zero_(zero)
        return self.flat_grad

    def grad_views(self, flat_grad):
        return [self._view(flat_grad, i) for i in range(len(self.params))]

    @torch.no_grad()
    def step(self, flat_grad=None, grad_scale=1.0):
        """``flat_grad``: gradients already packed in parameter order (e.g. by the all-reduce); default: pack
        the parameters' ``.grad``.  ``grad_scale`` multiplies the gradient inside the update (1/world_size after a
        sum all-reduce).  On the GPU the update is ONE launch of istnet_adam_step (include/istnet_optim.h)."""
        g = self.pack_grads() if flat_grad is None else flat_grad
        if not (self.flat.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.sync_lr()          # (inside a capture the fill would be recorded and replayed: sync before capturing)
        if self.flat.is_cuda:
            from . import _native
            if not g.is_contiguous() or g.dtype != torch.float32 or g.numel() != self.flat.numel():
                raise ValueError("FlatAdam.step: flat_grad must be a contiguous float32 tensor of the parameters' size")
            with torch.cuda.device(self.flat.device):
                # the launch advances step_count itself (last workgroup to finish): no `step_count += 1` kernel in front
                _native.check(_native.lib().istnet_adam_step_counting(
                    self.flat.numel(), self.flat.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(),
                    self.exp_avg_sq.data_ptr(), self.step_count.data_ptr(), self._ticket.data_ptr(),
                    self._lr_dev.data_ptr(), self._lr, self.betas[0], self.betas[1],
                    self.eps, self.weight_decay, float(grad_scale),
                    torch.cuda.current_stream(self.flat.device).cuda_stream), "adam_step_counting")
            # The native update writes the flat buffer through raw pointers, which autograd cannot see: tell it.  A
            # forward whose backward runs AFTER this step (two forwards then backward / step / backward, a prefetched
            # next forward) then raises "modified by an inplace operation" instead of silently differentiating with the
            # updated weights -- the fused nodes save views of live parameter memory (fused_mlp.FusedSALevelFunction).
            # (tests/test_optim.py::test_backward_after_native_step_raises pins the guard)
            torch.autograd.graph.increment_version(self.params)      # public API (torch >= 2.1); an absent one raises here
            return
        self.step_count += 1
        if grad_scale != 1.0:
            g = g * grad_scale
        # CPU (host-logic tests): the same update with plain ops
        b1, b2 = self.betas
        if self.weight_decay:
            g = g + self.weight_decay * self.flat
        self.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        t = float(self.step_count)
        denom = (self.exp_avg_sq.sqrt() / (1 - b2 ** t) ** 0.5).add_(self.eps)
        self.flat.addcdiv_(self.exp_avg, denom, value=-self.lr / (1 - b1 ** t))

    def state_dict(self):
        """The layout of ``torch.optim.Adam.state_dict()`` (what the reference's checkpoints hold, utils/solver.py
        save_checkpoint): per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` -- copies of this parameter's slice of the
        flat moment buffers (a snapshot: torch's ``load_state_dict`` keeps the tensors it is given when dtype and device
        already match) -- plus ``param_groups``.  Weights are NOT part of an optimizer state."""
        state = {}
        step_cpu = self.step_count.detach().to("cpu", torch.float32)    # torch.optim.Adam (non-capturable): CPU float tensor
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            state[i] = {"step": step_cpu.clone(),
                        "exp_avg": self._view(self.exp_avg, i).clone(),
                        "exp_avg_sq": self._view(self.exp_avg_sq, i).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": False}
        # keys a scheduler added to the group (CyclicLR / LambdaLR write ``initial_lr``, ``base_momentum`` ...): a resumed
        # scheduler built with last_epoch != -1 needs them back
        for k, v in self.param_groups[0].items():
            if k not in group and k != "params":
                group[k] = v
        group["params"] = list(range(len(self.params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts ``torch.optim.Adam.state_dict()`` of an optimizer over the same parameters in the same order (and
        this class's own).  Moments are copied into the flat buffers; the model's weights are left alone."""
        groups = sd["param_groups"]
        if len(groups) != 1:
            raise ValueError("FlatAdam.load_state_dict: exactly one parameter group expected")
        g = groups[0]
        keys = list(g["params"])
        if len(keys) != len(self.params):
            # A reference checkpoint of the non-frozen configuration was saved by Adam(model.parameters()) over ALL
            # parameters (train.py:101-107), while this optimizer holds the trainable ones only.  ``all_params`` (set by the
            # caller: ``opt.all_params = list(model.parameters())``) maps the checkpoint's indices over every parameter and
            # the non-trainable ones are skipped.
            every = getattr(self, "all_params", None)
            if every is not None and len(keys) == len(every):
                mine = {id(p) for p in self.params}
                keys = [k for k, p in zip(keys, every) if id(p) in mine]
            if len(keys) != len(self.params):
                raise ValueError(f"FlatAdam.load_state_dict: {len(g['params'])} parameters in the state, "
                                 f"{len(self.params)} in the optimizer (set opt.all_params = list(model.parameters()) to "
                                 "load a checkpoint saved over all parameters)")
        if g.get("amsgrad", False) or g.get("maximize", False):
            raise ValueError("FlatAdam.load_state_dict: amsgrad / maximize states are not supported")
        steps = set()
        for i, (key, p, o) in enumerate(zip(keys, self.params, self.offsets)):
            st = sd["state"].get(key)
            n = p.numel()
            if st is None:                       # parameter never stepped: zero moments
                self.exp_avg[o:o + n].zero_(); self.exp_avg_sq[o:o + n].zero_()
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"FlatAdam.load_state_dict: shape mismatch for parameter {key}: "
                                 f"{tuple(st['exp_avg'].shape)} vs {tuple(p.shape)}")
            self._view(self.exp_avg, i).copy_(st["exp_avg"])
            self._view(self.exp_avg_sq, i).copy_(st["exp_avg_sq"])
            steps.add(float(st["step"]))
        if len(steps) > 1:
            raise ValueError("FlatAdam.load_state_dict: parameters with different step counts cannot share one flat step")
        self.step_count.fill_(steps.pop() if steps else 0.0)
        self.lr = float(g["lr"])
        self.betas, self.eps, self.weight_decay = tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"])
        self.param_groups[0].update(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay)
        for k, v in g.items():                   # scheduler bookkeeping (initial_lr, ...) travels with the group
            if k not in self.param_groups[0] and k != "params":
                self.param_groups[0][k] = v

"""Train-step harness of the SV models on MI355X.

What the reference's training loop does per batch (main_cls_dgcnn.py:165-185): optional z / SO(3) rotation of the
clouds, permute to [B,3,N], zero_grad, forward, utils.cal_loss, backward, optimizer.step; per epoch
CosineAnnealingLR.step (:128-135,187).  Here:

  cal_loss          utils.py:33-50 as one HIP kernel (label-smoothed cross entropy)
  TrainStep         fwd + cal_loss + bwd (+ the data-parallel gradient all-reduce) on fixed device buffers, launched eagerly or
                    replayed as ONE captured HIP graph (the step is ~200 short kernels: launch-bound when launched one by one)
  rotate_clouds     the per-batch augmentation (main_cls_dgcnn.py:168-178) without pytorch3d
  FlatAdam/FlatSGD  torch.optim.Adam / SGD semantics (main_cls_dgcnn.py:128-133) as ONE kernel over the flat parameter /
                    gradient buffers (svnet_amd/csrc/optim.hip) instead of one small kernel chain per parameter tensor
  CosineLR          torch.optim.lr_scheduler.CosineAnnealingLR closed form (main_cls_dgcnn.py:135)
"""
import math

import torch

from . import _ops, config
from .dist import GradBucket


def cal_loss(pred, target, smoothing=True):
    """Label-smoothed cross entropy (eps = 0.2), mean over rows.  pred [R,C] float32, target [R] int64."""
    return _ops.SmoothCE.apply(pred, target.contiguous().view(-1), 0.2 if smoothing else 0.0)


def seg_loss(pred, target):
    """cal_loss on the [B,num_part,N] logits of the part-segmentation models (main_partseg_dgcnn.py: rows = points)."""
    return cal_loss(pred.permute(0, 2, 1).reshape(-1, pred.shape[1]), target.reshape(-1))


class TrainStep:
    """One data-parallel training step of `model` on FIXED device buffers: gradients of all parameters end up in ONE flat
    bucket (`self.bucket.flat`, every p.grad a view into it), averaged over the ranks when torch.distributed is initialised.

        step = TrainStep(model, inputs=(x,), target=y)        # x, y: device tensors that are refilled in place per batch
        step.capture()                                         # optional: record fwd+loss+bwd once as a HIP graph
        loss = step.run()                                      # replay (or eager launch) + gradient all-reduce

    The loss tensor returned by run() lives in a fixed buffer when the step is captured (read it before the next run()).
    """

    def __init__(self, model, inputs, target, loss_fn=cal_loss):
        self.model, self.inputs, self.target, self.loss_fn = model, tuple(inputs), target, loss_fn
        self.bucket = GradBucket(model.parameters())
        self.graph = None
        self.loss = None
        self._stream = None
        self._arena = _ops._ZeroArena()       # this step's zero-filled scratch: a captured graph has its addresses baked in

    def _deferral_is_safe(self):
        """config.DEFER_WGRAD hands autograd parameter gradients that the side stream has not written yet; that is sound only while
        autograd merely ADOPTS them as .grad (never reads them on the main stream): every parameter used by ONE module only (a shared
        one gets two gradients, added on the main stream), .grad None at backward (bucket.begin()), no tensor / post-accumulate hooks
        (they run on the main stream).  Anything else falls back to the per-layer join.  What this static check cannot see - one
        module applied twice in a forward, a parameter read by two Functions - is counted at run time: _ops.DEFERRED.first_use() sends the
        second backward of the same weights down the joined schedule."""
        seen = set()
        for _, p in self.model.named_parameters(remove_duplicate=False):
            if id(p) in seen:
                return False
            seen.add(id(p))
            if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
                return False
        return True

    def fwd_bwd(self, planes_external=False):
        """zero_grad -> forward -> loss -> backward, gradients packed into the flat bucket (no optimizer step)."""
        _ops.begin_step(self.bucket.flat.device, planes_external, self._arena)   # one zero fill for the step's accumulators; stale packed weights rebuilt on the side stream
        try:
            self.bucket.begin()
            loss = self.loss_fn(self.model(*self.inputs), self.target)
            # nothing reads a parameter gradient before pack() (checked: _deferral_is_safe): see _ops._Deferred
            _ops.DEFERRED.active = bool(config.DEFER_WGRAD) and self._deferral_is_safe()
            if loss.dim() == 0 and loss.is_cuda and loss.dtype == torch.float32:
                torch.autograd.backward(loss, _ops.UNIT_GRAD.get(loss.device))      # (a cached 1.0: no fill, no multiplication by one)
            else:
                loss.backward()
            _ops.DEFERRED.join(self.bucket.flat.device)
            self.bucket.pack()                              # one batched copy of all gradients into the flat bucket
        finally:
            _ops.end_step()
        return loss.detach()

    def capture(self, warmup=2, before_capture=None):
        """Record fwd_bwd() into a HIP graph.  The eager warm-up steps (allocator warm-up) run on the SAME side stream that is
        then captured: autograd runs every backward node on the stream its forward ran on, so a warm-up on another stream
        whose autograd graph is still alive (AccumulateGrad nodes) would fork the capture onto that stream.
        `before_capture`: called between the warm-up and the capture (diagnostics: tools/step_clock.py arms its stamps there)."""
        dev = self.bucket.flat.device
        self._stream = torch.cuda.Stream(device=dev, priority=config.MAIN_PRIORITY)
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._stream):
            for _ in range(warmup):
                loss = self.fwd_bwd()
                del loss
        torch.cuda.current_stream(dev).wait_stream(self._stream)
        torch.cuda.synchronize(dev)
        if before_capture is not None:
            before_capture()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=self._stream):
            self.loss = self.fwd_bwd(planes_external=True)    # (the packing launches stay outside the graph: run() refreshes what is stale)
        self.graph = graph
        return self

    def run(self, all_reduce=True):
        if self.graph is not None:
            _ops.PLANES.refresh(self.bucket.flat.device)      # packed weight forms: only what an optimizer step (or a load) changed
            self.graph.replay()
            loss = self.loss
        else:
            loss = self.fwd_bwd()
        if all_reduce:
            self.bucket.all_reduce_mean()
        return loss


class ForwardStep:
    """Forward only, eval() + no_grad, on fixed device buffers (the evaluation loop of main_cls_dgcnn.py:218-251), eager or
    as a captured HIP graph.  `self.out` holds the logits of the last run()."""

    def __init__(self, model, inputs):
        self.model, self.inputs = model, tuple(inputs)
        self.graph = None
        self.out = None
        self._arena = _ops._ZeroArena()

    def forward(self, planes_external=False):
        was_training = self.model.training
        self.model.eval()
        _ops.begin_step(self.inputs[0].device, planes_external, self._arena)
        try:
            with torch.no_grad():
                return self.model(*self.inputs)
        finally:
            _ops.end_step()
            self.model.train(was_training)        # (a TrainStep on the same model must not silently train in eval mode)

    def capture(self, warmup=2):
        dev = self.inputs[0].device
        stream = torch.cuda.Stream(device=dev, priority=config.MAIN_PRIORITY)
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream):
            for _ in range(warmup):
                self.forward()
        torch.cuda.current_stream(dev).wait_stream(stream)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            self.out = self.forward(planes_external=True)
        self.graph = graph
        return self

    def run(self):
        if self.graph is not None:
            _ops.PLANES.refresh(self.inputs[0].device)
            self.graph.replay()
        else:
            self.out = self.forward()
        return self.out


# ----------------------------------------------------------------------------- augmentation

def rotate_clouds(x, mode, generator=None):
    """Per-cloud random rotation of x [B,3,N] on its device (main_cls_dgcnn.py:168-178: 'z' = RotateAxisAngle about Z with a
    uniform angle, 'so3' = a uniform random rotation per cloud; anything else = no rotation).  Returns a new [B,3,N] tensor."""
    if mode not in ("z", "so3"):
        return x
    B = x.shape[0]
    dev = x.device
    if mode == "z":
        a = torch.rand(B, generator=generator, device=dev) * (2.0 * math.pi)
        c, s = torch.cos(a), torch.sin(a)
        z, o = torch.zeros_like(a), torch.ones_like(a)
        R = torch.stack([c, -s, z, s, c, z, z, z, o], dim=1).view(B, 3, 3)
    else:
        q = torch.randn(B, 4, generator=generator, device=dev)            # uniform on S^3 -> uniform on SO(3)
        q = q / q.norm(dim=1, keepdim=True)
        w, i, j, k = q.unbind(1)
        R = torch.stack([1 - 2 * (j * j + k * k), 2 * (i * j - k * w), 2 * (i * k + j * w),
                         2 * (i * j + k * w), 1 - 2 * (i * i + k * k), 2 * (j * k - i * w),
                         2 * (i * k - j * w), 2 * (j * k + i * w), 1 - 2 * (i * i + j * j)], dim=1).view(B, 3, 3)
    return torch.bmm(R, x)


# ----------------------------------------------------------------------------- optimizers on flat buffers

def invalidate_packed_weights():
    """Tell the packed-weight cache (_ops._PlaneCache) that weights were changed in a way autograd's version counters do not see
    (a write through `.data`, a raw pointer, a custom kernel): everything is re-packed at the next step.  torch.optim steps,
    load_state_dict and in-place ops under no_grad ARE seen; the flat optimizers below call this themselves."""
    _ops.PLANES.invalidate()


class FlatParams:
    """Re-homes every trainable parameter of `model` into ONE flat fp32 buffer (each p.data becomes a view), in the
    order of model.parameters() — the same order as the GradBucket — so that an optimizer step is a single kernel."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)
                off += n


class _FlatOptimizer:
    def __init__(self, flat_params, bucket, lr):
        if flat_params.flat.numel() != bucket.flat.numel():
            raise ValueError("parameter and gradient buffers differ in size")
        self.p, self.g = flat_params.flat, bucket.flat
        self.shapes = [tuple(p.shape) for p in flat_params.params]
        self.lr = float(lr)
        self.base_lr = float(lr)
        self.steps = 0

    # state_dict / load_state_dict in torch.optim's layout (the reference saves optimizer.state_dict() into its checkpoints and
    # resumes with optimizer.load_state_dict, main_cls_dgcnn.py:147-148, utils.py:141-171): per-parameter state tensors keyed by
    # the parameter's index, one param_group - so a checkpoint written by either side resumes on the other.
    def _split(self, flat):
        out, off = [], 0
        for shp in self.shapes:
            n = 1
            for d in shp:
                n *= d
            out.append(flat[off:off + n].view(shp))
            off += n
        return out

    def _gather(self, state, key, flat):
        with torch.no_grad():
            for i, dst in enumerate(self._split(flat)):
                if i in state and key in state[i]:
                    dst.copy_(state[i][key].to(device=dst.device, dtype=dst.dtype).view_as(dst))
                else:
                    dst.zero_()


    # ---- the step as a captured graph.  An eager step is the update kernel plus, before the next forward, ~25 small launches that re-pack
    # the binarized weights it changed (_ops.PLANES.refresh): 0.18 ms per step of latency-bound launches behind a 4.4 ms replay.  capture()
    # records [update kernel -> every re-pack, spread over four streams] once; step() then costs one 32-byte copy of the step's scalars
    # (learning rate, bias corrections) and one graph launch.  The update kernel reads those scalars from device memory.
    def _hyper_values(self):
        raise NotImplementedError

    def _launch_dev(self):
        raise NotImplementedError

    def capture(self):
        """Call after the model has run at least once (the packed forms exist) and the parameters live in their flat buffer.
        The warm-up and the capture launch the REAL update kernel on the live buffers with neutral scalars (lr 0, betas 1): every
        product with a finite gradient is then exactly "keep the value", but 0 * inf = NaN would poison the moments - so a non-finite
        gradient bucket is refused here (one host sync, at capture time only).  The graph bakes the buffers' addresses in:
        they are recorded, and a step after they moved (re-homed parameters, a load that re-allocated the state) raises."""
        dev = self.p.device
        if not bool(torch.isfinite(self.g).all()):
            raise FloatingPointError("FlatOptimizer.capture(): the gradient bucket holds inf / NaN - capture after a finite backward "
                                     "(the neutral warm-up step would write NaN into the optimizer state)")
        self._hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        # (a ring of pinned staging slots: the host runs ahead of the device, so a slot is rewritten only after the copy that read it last)
        self._hyper_ring = [(torch.zeros(8, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(16)]
        self._hyper_slot = 0
        self._helpers = [torch.cuda.Stream(device=dev) for _ in range(4)]
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))

        def record():
            self._launch_dev()
            main = torch.cuda.current_stream(dev)
            issued = [0]

            def run(rebuild):
                h = self._helpers[issued[0] % len(self._helpers)]
                if issued[0] < len(self._helpers):
                    h.wait_stream(main)                       # fork: behind the update kernel
                issued[0] += 1
                with torch.cuda.stream(h):
                    rebuild()
            keys = _ops.PLANES.rebuild_all(run)
            for h in self._helpers[:issued[0]]:
                main.wait_stream(h)                           # join
            return keys
        # (nothing is stepped here: the warm-up and the capture run with a zero learning rate and neutral bias corrections)
        with torch.cuda.stream(stream):
            self._put_hyper(self._neutral_hyper())
            record()
        torch.cuda.current_stream(dev).wait_stream(stream)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            self._keys = record()
        self._graph = graph
        self._captured_ptrs = self._buffer_ptrs()
        return self

    def _buffer_ptrs(self):
        return tuple(t.data_ptr() for t in self._state_tensors())

    def _state_tensors(self):
        raise NotImplementedError

    def _put_hyper(self, vals):
        host, ev = self._hyper_ring[self._hyper_slot]
        self._hyper_slot = (self._hyper_slot + 1) % len(self._hyper_ring)
        ev.synchronize()                                      # (returns at once unless the device is 16 steps behind)
        host[:len(vals)] = torch.tensor(vals, dtype=torch.float32)
        self._hyper.copy_(host, non_blocking=True)
        ev.record()

    def _step_captured(self):
        if self._buffer_ptrs() != self._captured_ptrs:
            raise RuntimeError("FlatOptimizer: the parameter / gradient / state buffers moved since capture() (re-homed parameters or a "
                               "re-allocated state): the captured graph would update the old memory - call capture() again")
        self._put_hyper(self._hyper_values())
        self._graph.replay()
        if set(_ops.PLANES.entries) <= set(self._keys):
            _ops.PLANES.mark_fresh(self._keys)                # everything that exists was re-packed inside the graph
        else:
            _ops.PLANES.invalidate()                          # a packed form created after capture(): the eager refresh takes over


class FlatAdam(_FlatOptimizer):
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay) — L2 weight decay added to the gradient, bias-corrected
    moments, denominator sqrt(v_hat) + eps (main_cls_dgcnn.py:132-133, the optimizer of the binary models)."""

    def __init__(self, flat_params, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(flat_params, bucket, lr)
        self.b1, self.b2, self.eps, self.wd = float(betas[0]), float(betas[1]), float(eps), float(weight_decay)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)

    def state_dict(self):
        m, v = self._split(self.m), self._split(self.v)
        state = {} if self.steps == 0 else {i: {"step": torch.tensor(float(self.steps)), "exp_avg": m[i].detach().clone(),
                                                "exp_avg_sq": v[i].detach().clone()} for i in range(len(self.shapes))}
        group = {"lr": self.lr, "betas": (self.b1, self.b2), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "initial_lr": self.base_lr,
                 "params": list(range(len(self.shapes)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        if len(g["params"]) != len(self.shapes):
            raise ValueError("optimizer state has %d parameters, the model has %d" % (len(g["params"]), len(self.shapes)))
        if g.get("amsgrad"):
            raise NotImplementedError("FlatAdam: amsgrad state cannot be loaded")
        self.lr, (self.b1, self.b2), self.eps, self.wd = float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"]), float(g["weight_decay"])
        self.base_lr = float(g.get("initial_lr", self.base_lr))
        state = {int(k): v for k, v in sd["state"].items()}
        steps = {int(float(st["step"])) for st in state.values()}
        if len(steps) > 1:
            raise NotImplementedError("FlatAdam: per-parameter step counts differ (%r)" % sorted(steps))
        self.steps = steps.pop() if steps else 0
        self._gather(state, "exp_avg", self.m)
        self._gather(state, "exp_avg_sq", self.v)

    def _neutral_hyper(self):
        return [0.0, 1.0, 1.0, self.eps, 0.0, 1.0, 1.0]      # lr 0, betas 1: parameters and both moments keep their values

    def _state_tensors(self):
        return (self.p, self.g, self.m, self.v)

    def _hyper_values(self):
        bc1 = 1.0 - self.b1 ** self.steps
        bc2 = 1.0 - self.b2 ** self.steps
        return [self.lr, self.b1, self.b2, self.eps, self.wd, bc1, 1.0 / math.sqrt(bc2)]

    def _launch_dev(self):
        _ops.call("svnet_adam_step_dev_f32", _ops._p(self.p), _ops._p(self.g), _ops._p(self.m), _ops._p(self.v), self.p.numel(),
                  _ops._p(self._hyper), _ops._stream())

    def step(self):
        self.steps += 1
        if getattr(self, "_graph", None) is not None:
            return self._step_captured()
        _ops.PLANES.invalidate()                              # (the kernel writes the flat buffer behind autograd's version counters)
        _ops.call("svnet_adam_step_f32", _ops._p(self.p), _ops._p(self.g), _ops._p(self.m), _ops._p(self.v), self.p.numel(),
                  self.lr, self.b1, self.b2, self.eps, self.wd, self.steps, _ops._stream())


class FlatSGD(_FlatOptimizer):
    """torch.optim.SGD(lr, momentum, weight_decay) (main_cls_dgcnn.py:129-130, the optimizer of the fp models):
    g += wd*p; buf = g on the first step, momentum*buf + g afterwards; p -= lr*buf."""

    def __init__(self, flat_params, bucket, lr=0.1, momentum=0.9, weight_decay=0.0):
        super().__init__(flat_params, bucket, lr)
        self.momentum, self.wd = float(momentum), float(weight_decay)
        self.buf = torch.zeros_like(self.p)

    def state_dict(self):
        buf = self._split(self.buf)
        state = {} if self.steps == 0 or self.momentum == 0.0 else {i: {"momentum_buffer": buf[i].detach().clone()} for i in range(len(self.shapes))}
        group = {"lr": self.lr, "momentum": self.momentum, "dampening": 0, "weight_decay": self.wd, "nesterov": False, "maximize": False,
                 "foreach": None, "differentiable": False, "fused": None, "initial_lr": self.base_lr, "params": list(range(len(self.shapes)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        if len(g["params"]) != len(self.shapes):
            raise ValueError("optimizer state has %d parameters, the model has %d" % (len(g["params"]), len(self.shapes)))
        if g.get("nesterov") or g.get("dampening", 0) != 0:
            raise NotImplementedError("FlatSGD: nesterov / dampening are not supported")
        self.lr, self.momentum, self.wd = float(g["lr"]), float(g["momentum"]), float(g["weight_decay"])
        self.base_lr = float(g.get("initial_lr", self.base_lr))
        state = {int(k): v for k, v in sd["state"].items()}
        has = [i for i in state if state[i].get("momentum_buffer") is not None]
        self._gather(state, "momentum_buffer", self.buf)
        self.steps = 1 if has else 0          # (torch.optim.SGD keeps no step count: a present buffer means "not the first step")

    def _neutral_hyper(self):
        return [0.0, 0.0, 0.0, 1.0]

    def _state_tensors(self):
        return (self.p, self.g, self.buf)

    def _hyper_values(self):
        return [self.lr, self.momentum, self.wd, 1.0 if self.steps == 1 else 0.0]

    def _launch_dev(self):
        _ops.call("svnet_sgd_step_dev_f32", _ops._p(self.p), _ops._p(self.g), _ops._p(self.buf), self.p.numel(), _ops._p(self._hyper),
                  _ops._stream())

    def step(self):
        self.steps += 1
        if getattr(self, "_graph", None) is not None:
            return self._step_captured()
        _ops.PLANES.invalidate()
        _ops.call("svnet_sgd_step_f32", _ops._p(self.p), _ops._p(self.g), _ops._p(self.buf), self.p.numel(), self.lr, self.momentum,
                  self.wd, int(self.steps == 1), _ops._stream())


class CosineLR:
    """CosineAnnealingLR(optimizer, T_max, eta_min), stepped once per epoch (main_cls_dgcnn.py:135,187), closed form:
    lr(e) = eta_min + (base - eta_min) * (1 + cos(pi * e / T_max)) / 2."""

    def __init__(self, optimizer, T_max, eta_min=0.0):
        self.opt, self.T_max, self.eta_min, self.epoch = optimizer, int(T_max), float(eta_min), 0

    def state_dict(self):
        """torch.optim.lr_scheduler.CosineAnnealingLR.state_dict()'s keys (minus the optimizer reference)."""
        return {"T_max": self.T_max, "eta_min": self.eta_min, "base_lrs": [self.opt.base_lr], "last_epoch": self.epoch,
                "_step_count": self.epoch + 1, "_last_lr": [self.opt.lr]}

    def load_state_dict(self, sd):
        self.T_max, self.eta_min, self.epoch = int(sd["T_max"]), float(sd["eta_min"]), int(sd["last_epoch"])
        self.opt.base_lr = float(sd["base_lrs"][0])
        self.opt.lr = self.eta_min + (self.opt.base_lr - self.eta_min) * (1.0 + math.cos(math.pi * self.epoch / self.T_max)) / 2.0

    def step(self):
        self.epoch += 1
        self.opt.lr = self.eta_min + (self.opt.base_lr - self.eta_min) * (1.0 + math.cos(math.pi * self.epoch / self.T_max)) / 2.0
        return self.opt.lr

"""HIP-kernel-backed stand-ins for the three torch objects DreamNetwork hands to callers:
``torch.nn.MSELoss`` (dream/network.py:260-261), ``torch.optim.Adam`` and ``torch.optim.SGD``
(dream/network.py:666-685, PyTorch defaults: betas (0.9, 0.999), eps 1e-8, no weight decay)."""
import torch

from . import ops


class _MSEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, target):
        loss, grad = ops.mse_fwd_bwd(out.detach(), target.detach(), want_grad=out.requires_grad, kind="mse")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        ctx.grad = None
        return (grad * g if grad is not None else None), None


class HipMSELoss(torch.nn.Module):
    """mean((input - target)^2); the gradient 2(o-t)/N is produced by the same kernel pass."""

    def forward(self, input, target):
        return _MSEFunction.apply(input, target)


class _SmoothL1Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, target):
        loss, grad = ops.mse_fwd_bwd(out.detach(), target.detach(), want_grad=out.requires_grad, kind="huber")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        ctx.grad = None
        return (grad * g if grad is not None else None), None


class HipSmoothL1Loss(torch.nn.Module):
    """torch.nn.SmoothL1Loss() (beta 1, mean) -- the reference's "huber" loss type (dream/network.py:262-263)."""

    def forward(self, input, target):
        return _SmoothL1Function.apply(input, target)


def _flat_span(tensors):
    """If every tensor is a contiguous fp32 view into ONE storage: (flat view covering min..max extent, float offsets of the
    tensors inside it); else None.  The parameters of a DreamNetwork model are such views (dream_amd/data_parallel.py:
    flatten_module_), and so are the gradients the data-parallel autograd node hands out."""
    if not tensors:
        return None
    t0 = tensors[0]
    st = t0.untyped_storage()
    base = st.data_ptr()
    lo, hi, offs = None, None, []
    for t in tensors:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.untyped_storage().data_ptr() != base or t.device != t0.device:
            return None
        o = (t.data_ptr() - base) // 4
        offs.append(o)
        lo = o if lo is None else min(lo, o)
        hi = o + t.numel() if hi is None else max(hi, o + t.numel())
    flat = torch.empty(0, dtype=torch.float32, device=t0.device).set_(st, lo, (hi - lo,), (1,))
    return flat, [o - lo for o in offs]


def attach_data_parallel(optimizer, dp):
    """DreamNetwork.enable_training: the optimizer of a single-process data-parallel network (dream_amd/data_parallel.py) applies
    every step to the replicas as well (DreamDataParallel.step_replicas) instead of having them re-copied from the master."""
    optimizer._dp = dp if hasattr(dp, "step_replicas") else None


class _FlatStepMixin:
    """One kernel launch per optimizer step instead of one per tensor (46 for vgg_q, 318 for resnet_h): parameters that
    are views of one flat buffer are updated as that buffer; the gradients are used in place when they form the same layout
    (data-parallel path) or gathered into a flat buffer by one multi-tensor copy."""

    def _flat_plan(self, params):
        key = tuple((p.data_ptr(), p.numel()) for p in params)
        plan = getattr(self, "_plan", None)
        if plan is None or plan["key"] != key:
            span = _flat_span([p.data for p in params])
            plan = {"key": key, "span": span}
            if span is not None:
                flat, offs = span
                plan["grad"] = torch.zeros_like(flat)
                plan["grad_views"] = [plan["grad"][o:o + p.numel()].view(p.shape) for o, p in zip(offs, params)]
            self._plan = plan
        return plan

    def _flat_grad(self, plan, params):
        grads = [p.grad for p in params]
        gspan = _flat_span(grads)
        if gspan is not None and gspan[1] == plan["span"][1] and gspan[0].numel() == plan["span"][0].numel():
            return gspan[0]                                   # already laid out like the parameters: use in place
        # gather into the flat buffer (padding between tensors stays zero): ONE launch of our own when the gradients are plain
        # contiguous fp32 tensors (torch._foreach_copy_ issues one hipMemcpyAsync per tensor: 318 of them for resnet_h, 1.4 ms)
        mc = plan.get("multi_copy")
        if mc is None:
            try:                                                # (CPU tensors outside the test emulator: ptr() refuses them)
                mc = plan["multi_copy"] = ops.MultiCopyPlan(plan["grad_views"])
            except RuntimeError:
                mc = plan["multi_copy"] = False
        if mc and mc.matches(grads):
            mc.run(grads)
        else:
            torch._foreach_copy_(plan["grad_views"], grads)
        return plan["grad"]


class HipAdam(_FlatStepMixin, torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    def load_state_dict(self, state_dict):
        """torch.optim.Optimizer.load_state_dict hands every parameter CLONES of the saved moments; the flat plan (whose
        moment buffers are the source of truth of the one-launch step) is dropped so that the next step adopts them."""
        super().load_state_dict(state_dict)
        self._plan = None
        dp = getattr(self, "_dp", None)
        if dp is not None:
            dp._opt_state.clear()                          # the replicas' moment buffers belonged to the state just replaced

    def _adopt_moments(self, plan, params, offs):
        """Build the flat moment buffers of ``plan``; per-parameter state that already exists (a resumed optimizer:
        load_state_dict, or a re-plan after the parameters moved) is copied into them before the state is re-pointed at the
        views.  -> False when the existing per-parameter step counts disagree (the per-tensor path keeps them apart)."""
        flat = plan["span"][0]
        steps = {int(self.state[p]["step"]) for p in params if self.state[p]}
        if len(steps) > 1 or (steps and any(not self.state[p] for p in params)):
            return False
        m, v = torch.zeros_like(flat), torch.zeros_like(flat)
        for p, o in zip(params, offs):
            st = self.state[p]
            mv, vv = m[o:o + p.numel()].view(p.shape), v[o:o + p.numel()].view(p.shape)
            if st:
                mv.copy_(st["exp_avg"])
                vv.copy_(st["exp_avg_sq"])
            st["exp_avg"], st["exp_avg_sq"], st["flat"] = mv, vv, True
            st.setdefault("step", 0)
        plan["moments"] = (m, v)
        plan["step"] = steps.pop() if steps else 0
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            params = group["params"]
            plan = self._flat_plan(params) if len(self.param_groups) == 1 and all(p.grad is not None for p in params) else None
            if plan is not None and plan["span"] is not None and "moments" not in plan and not plan.get("per_tensor"):
                if not self._adopt_moments(plan, params, plan["span"][1]):
                    plan["per_tensor"] = True
            if plan is not None and plan["span"] is not None and "moments" in plan:
                flat, offs = plan["span"]
                plan["step"] += 1
                grad = self._flat_grad(plan, params)
                dp = getattr(self, "_dp", None)
                replicas_stepped = dp is not None and grad is not plan["grad"] and dp.step_replicas(
                    "adam", flat, grad, dict(lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"], step=plan["step"]), plan["moments"])
                ops.adam_step_(flat, grad, plan["moments"][0], plan["moments"][1], group["lr"], b1, b2, group["eps"], plan["step"])
                for p in params:
                    self.state[p]["step"] = plan["step"]
                    _bump_version(p)
                if replicas_stepped:
                    dp.mark_params_synced()
                continue
            for p in params:
                if p.grad is None:
                    continue
                st = self._state_of(p)
                st["step"] = int(st["step"]) + 1
                ops.adam_step_(p.data, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2,
                               group["eps"], st["step"])
                _bump_version(p)
        return loss


class HipSGD(_FlatStepMixin, torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3):
        super().__init__(params, dict(lr=lr))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            params = group["params"]
            plan = self._flat_plan(params) if len(self.param_groups) == 1 and all(p.grad is not None for p in params) else None
            if plan is not None and plan["span"] is not None:
                grad = self._flat_grad(plan, params)
                dp = getattr(self, "_dp", None)
                replicas_stepped = dp is not None and grad is not plan["grad"] and dp.step_replicas(
                    "sgd", plan["span"][0], grad, dict(lr=group["lr"]), None)
                ops.sgd_step_(plan["span"][0], grad, group["lr"])
                for p in params:
                    _bump_version(p)
                if replicas_stepped:
                    dp.mark_params_synced()
                continue
            for p in params:
                if p.grad is not None:
                    ops.sgd_step_(p.data, p.grad.contiguous(), group["lr"])
                    _bump_version(p)
        return loss


def _bump_version(p):
    """The kernels write through raw pointers; tell autograd / the packed-weight cache that the parameter changed."""
    ops.bump_version(p)

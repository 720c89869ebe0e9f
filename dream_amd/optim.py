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


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                ops.adam_step_(p.data, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2,
                               group["eps"], st["step"])
                _bump_version(p)
        return loss


class HipSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3):
        super().__init__(params, dict(lr=lr))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    ops.sgd_step_(p.data, p.grad.contiguous(), group["lr"])
                    _bump_version(p)
        return loss


def _bump_version(p):
    """The kernels write through raw pointers; tell autograd / the packed-weight cache that the parameter changed."""
    ops.bump_version(p)

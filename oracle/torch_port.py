"""Op-for-op torch-CPU restatement of the reference training step — TEST/BENCH INFRASTRUCTURE ONLY
(only tests/ and bench.py's cpu_baseline leg import it).

It issues the same ATen CPU kernels the reference's autograd + torch.optim.Adam path dispatches per
step (3 index, sub, mul, sum, sigmoid, BCE, 3 dense zero-fills + index_put_(accumulate),
the dense add of V's two branches, and the single-tensor Adam op chain), so its step time is a faithful
stand-in for the reference's own CPU path on a box the reference cannot travel to.
Follows structure.py:845-852 and torch/optim/adam.py `_single_tensor_adam`.
"""
import torch
import torch.nn.functional as F


def train_steps(U, V, state, u, i, j, z, B, step0, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, wd=1e-5):
    """In place on fp32 CPU tensors U, V and state = dict(mU, vU, mV, vV).  u,i,j int64, z float64 (as collated).
    Returns the list of per-step losses (Python floats, one .item() per step as at structure.py:852)."""
    b1, b2 = betas
    N = u.numel()
    losses = []
    step = step0
    for off in range(0, N, B):
        ub, ib, jb = u[off:off + B], i[off:off + B], j[off:off + B]
        zb = z[off:off + B].float()
        ue, ie, je = U[ub], V[ib], V[jb]                         # aten::index x3
        diff = ie - je
        p = torch.sigmoid(torch.sum(ue * diff, dim=1))
        loss = F.binary_cross_entropy(p, zb)
        nb = zb.numel()
        a = (p - zb) / torch.clamp((1 - p) * p, min=1e-12) / nb  # binary_cross_entropy_backward
        g = (a * (1 - p) * p).unsqueeze(1)                       # sigmoid_backward
        gu = g * ue
        dU = torch.zeros_like(U).index_put_((ub,), g * diff, accumulate=True)
        dV = torch.zeros_like(V).index_put_((ib,), gu, accumulate=True)
        dV.add_(torch.zeros_like(V).index_put_((jb,), -gu, accumulate=True))
        step += 1
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        for prm, grad, m, v in ((U, dU, state["mU"], state["vU"]), (V, dV, state["mV"], state["vV"])):
            if wd != 0:
                grad = grad.add(prm, alpha=wd)
            m.lerp_(grad, 1 - b1)
            v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
            denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
            prm.addcdiv_(m, denom, value=-(lr / bc1))
        losses.append(loss.item())
    return losses

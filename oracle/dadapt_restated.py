"""TEST INFRASTRUCTURE ONLY -- restatement of ``dadaptation.DAdaptAdam`` (dadaptation==3.2).

PARITY UNPINNED: the reference pins ``dadaptation == 3.2`` (``pyproject.toml:12``) and calls it at
``vamb/encode.py:8,578`` as ``DAdaptAdam(self.parameters(), decouple=True)``.  The package is not
vendored under ``/root/reference`` and is not installed in this image (no network), so this file
restates the published algorithm (Defazio & Mishchenko, "Learning-Rate-Free Learning by
D-Adaptation", ICML 2023, Algorithm "D-Adapted Adam", as shipped in facebookresearch/dadaptation
``dadapt_adam.py``) for the constructor arguments Vamb uses.  The only reference tests that touch it
are behavioural (``test/test_encode.py:152-168`` "loss falls", ``test/test_results.py:87-100`` "runs").

Defaults implied by Vamb's call: lr=1.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0 (so
``decouple=True`` is a no-op), use_bias_correction=False, d0=1e-6, growth_rate=inf, no FSDP.

Per step, with group state (d, k, numerator_weighted) and per-tensor state (s, exp_avg, exp_avg_sq):

    dlr = d * lr
    for each tensor with a grad g:
        denom          = sqrt(exp_avg_sq) + eps                     # OLD exp_avg_sq
        numerator_acum += dlr * <g, s / denom>                      # fp32 dot, accumulated in fp64
        exp_avg        = b1 * exp_avg + dlr (1 - b1) g
        exp_avg_sq     = b2 * exp_avg_sq + (1 - b2) g^2
        s              = sqrt(b2) s + dlr (1 - sqrt(b2)) g
        sk_l1         += |s|_1                                      # fp32 sum, accumulated in fp64
    numerator_weighted = sqrt(b2) numerator_weighted + (1 - sqrt(b2)) numerator_acum
    if sk_l1 == 0: return                                          # nothing stored, k not advanced
    d_hat = numerator_weighted / ((1 - sqrt(b2)) sk_l1)
    d     = max(d, min(d_hat, d * growth_rate))
    for each tensor: p -= exp_avg / (sqrt(exp_avg_sq) + eps)        # NEW moments; dlr already inside exp_avg
    k += 1
"""
from __future__ import annotations

import math

import torch


class DAdaptAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 decouple=False, d0=1e-6, growth_rate=float("inf")):
        if weight_decay != 0.0:
            raise NotImplementedError("Vamb never passes weight_decay (encode.py:578)")
        defaults = dict(lr=lr, betas=betas, eps=eps, d=d0, k=0, numerator_weighted=0.0,
                        growth_rate=growth_rate, decouple=decouple)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None if closure is None else closure()
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        d = g0["d"]
        lr = max(g["lr"] for g in self.param_groups)
        dlr = d * lr
        sqrt_b2 = b2 ** 0.5
        numerator_weighted = g0["numerator_weighted"]
        sk_l1 = 0.0
        numerator_acum = 0.0

        for group in self.param_groups:
            eps = group["eps"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["s"] = torch.zeros_like(p)
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                s, m, v = st["s"], st["exp_avg"], st["exp_avg_sq"]
                denom = v.sqrt().add_(eps)
                numerator_acum += dlr * torch.dot(g.flatten(), s.div(denom).flatten()).item()
                m.mul_(b1).add_(g, alpha=dlr * (1 - b1))
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                s.mul_(sqrt_b2).add_(g, alpha=dlr * (1 - sqrt_b2))
                sk_l1 += s.abs().sum().item()

        numerator_weighted = sqrt_b2 * numerator_weighted + (1 - sqrt_b2) * numerator_acum
        if sk_l1 == 0:
            return loss
        if lr > 0.0:
            d_hat = numerator_weighted / ((1 - sqrt_b2) * sk_l1)
            d = max(d, min(d_hat, d * g0["growth_rate"]))

        for group in self.param_groups:
            group["numerator_weighted"] = numerator_weighted
            group["d"] = d
            eps = group["eps"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                st["step"] += 1
                denom = st["exp_avg_sq"].sqrt().add_(eps)
                p.addcdiv_(st["exp_avg"], denom, value=-1.0)
            group["k"] = group["k"] + 1
        return loss


def _selfcheck():  # pragma: no cover
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(8, 4))
    opt = DAdaptAdam([w])
    for _ in range(50):
        opt.zero_grad()
        ((w ** 2).sum()).backward()
        opt.step()
    assert math.isfinite(opt.param_groups[0]["d"]) and opt.param_groups[0]["d"] > 1e-6


if __name__ == "__main__":  # pragma: no cover
    _selfcheck()
    print("ok")

"""Energy minimisation on the HIP force kernels (module name as in the reference: chiron/minimze.py).

The reference hands ``jax.value_and_grad(potential_fn)`` to ``jaxopt.GradientDescent``
(chiron/minimze.py:5-45): accelerated gradient descent with a backtracking line search,
``tol = 1e-3`` on the gradient norm, returning an object whose ``.params`` are the minimised
coordinates.  There is no autodiff here: ``potential_fn`` must be the bound
``compute_energy`` of a potential that also offers ``compute_energy_and_force`` (the
analytic gradient the reference's own tests verify, chiron/potential.py:302-332); energies
and forces come from the same kernels as every other caller, the vector updates are plain
device tensor arithmetic.  jaxopt's exact iterate sequence is not reproduced (its version is
unpinned); the reference's tests only pin the minimum that is reached
(tests/test_minization.py:83-128).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Callable

import torch

from ._device import as_tensor


@dataclass
class MinimizerState:
    iter_num: int
    stepsize: float
    error: float  # gradient norm at the last accepted point
    value: float  # potential energy at ``params``


@dataclass
class OptStep:
    """Mirror of ``jaxopt.OptStep``: ``result.params`` are the minimised coordinates."""

    params: torch.Tensor
    state: MinimizerState

    def __iter__(self):
        return iter((self.params, self.state))


def _value_and_grad(potential_fn: Callable, nbr_list) -> Callable[[torch.Tensor], Any]:
    owner = getattr(potential_fn, "__self__", None)
    if owner is None or not hasattr(owner, "compute_energy_and_force"):
        raise TypeError(
            "minimize_energy needs the bound compute_energy of a potential with an analytic force "
            "(compute_energy_and_force); arbitrary callables cannot be differentiated here"
        )

    def fn(x):
        e, f = owner.compute_energy_and_force(x, nbr_list)
        return float(e), f  # gradient = -f

    return fn


def minimize_energy(coordinates, potential_fn: Callable, nbr_list=None, maxiter: int = 1000,
                    tol: float = 1e-3, maxls: int = 15, decrease_factor: float = 0.5) -> OptStep:
    """Minimise ``potential_fn(x[, nbr_list])`` starting from ``coordinates`` (chiron/minimze.py:5-45).

    Accelerated gradient descent (FISTA momentum) with a backtracking line search on the
    sufficient-decrease condition ``f(y - s g) <= f(y) - s/2 |g|^2``; stops after ``maxiter``
    iterations or when ``|g| <= tol``.  The neighbor list is used as given (the reference does
    not rebuild it during minimisation either).
    """
    value_and_grad = _value_and_grad(potential_fn, nbr_list)
    x = as_tensor(coordinates).clone()
    y = x.clone()
    t = 1.0
    stepsize = 1.0
    value, force = value_and_grad(x) if maxiter > 0 else (float("nan"), None)
    error = float("inf")
    it = 0
    f_y, force_y = value, force
    while it < maxiter:
        g2 = float((force_y * force_y).sum())
        error = math.sqrt(g2)
        if not math.isfinite(error):
            break
        if error <= tol:
            break
        # backtracking line search from a slightly enlarged previous step
        s = stepsize / decrease_factor
        accepted = False
        for _ in range(maxls + 1):
            cand = y + s * force_y  # y - s * grad
            f_c, force_c = value_and_grad(cand)
            if math.isfinite(f_c) and f_c <= f_y - 0.5 * s * g2:
                accepted = True
                break
            s *= decrease_factor
        if not accepted:
            # no sufficient decrease within maxls halvings: keep the shrunken step for the next
            # iteration (steep starting structures need many halvings) and drop the momentum
            stepsize = s
            it += 1
            if s < 1e-30:
                break
            if t > 1.0:
                y, t = x.clone(), 1.0
                f_y, force_y = value, force
            continue
        stepsize = s
        t_next = 0.5 * (1.0 + math.sqrt(1.0 + 4.0 * t * t))
        y_next = cand + ((t - 1.0) / t_next) * (cand - x)
        x, value, force = cand, f_c, force_c
        if t > 1.0:
            f_y, force_y = value_and_grad(y_next)
            if not math.isfinite(f_y) or f_y > value:  # momentum overshoot: restart (keeps descent monotone)
                y_next, t_next = x.clone(), 1.0
                f_y, force_y = f_c, force_c
        else:
            f_y, force_y = f_c, force_c
        y, t = y_next, t_next
        it += 1
    if maxiter > 0 and not math.isfinite(value):
        value = float("nan")
    return OptStep(params=x, state=MinimizerState(iter_num=it, stepsize=stepsize, error=error, value=value))

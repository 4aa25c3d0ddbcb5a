"""Dormand-Prince RK45 with the state resident on the device.

The reference integrates the probability-flow ODE with scipy.integrate.solve_ivp(method='RK45') on the host
(ldm/notebook_utils.py:345-358): every function evaluation moves the whole batch device -> host float64 -> device.
Here the float64 state, the seven fp32 stage derivatives and the error norm stay in HBM (mulan_rk_* entry points);
the host keeps only the step-size controller, i.e. one scalar read-back per step instead of six tensor round trips.
The controller restates scipy's (RK45 / OdeSolver / select_initial_step of scipy.integrate._ivp, the solver the
reference calls): same tableau, same RMS error norm, same accept / reject rule, so the step sequence is the one
solve_ivp would take.
"""
import math

import numpy as np
import torch

from . import ops

# Dormand & Prince 1980, the 5(4) pair with FSAL
C = (0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0)
A = ((),
     (1 / 5,),
     (3 / 40, 9 / 40),
     (44 / 45, -56 / 15, 32 / 9),
     (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
     (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656))
B = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84)
E = (-71 / 57600, 0.0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40)
SAFETY, MIN_FACTOR, MAX_FACTOR = 0.9, 0.2, 10.0
ORDER = 4                                   # error estimator order of RK45
EXPONENT = -1.0 / (ORDER + 1)


class OdeResult:
    def __init__(self, y, t, nfev, steps, rejected):
        self.y, self.t, self.nfev, self.steps, self.rejected = y, t, nfev, steps, rejected


def solve_fixed(fun, y0, t_grid):
    """the same Dormand-Prince steps on a prescribed time grid, no error control (parity tests: both sides take
    identical steps, which isolates the function evaluations and the state plumbing from the controller)"""
    n = y0.numel()
    dev = y0.device
    y = y0.detach().clone().contiguous()
    ynew = torch.empty_like(y)
    y32 = torch.empty(n, device=dev, dtype=torch.float32)
    K = torch.empty((7, n), device=dev, dtype=torch.float32)
    ops.rk_combine(y, K, (), 0.0, out32=y32)
    fun(float(t_grid[0]), y32, K[0])
    nfev = 1
    for t, t_new in zip(t_grid[:-1], t_grid[1:]):
        t, h = float(t), float(t_new) - float(t)
        for stage in range(1, 6):
            ops.rk_combine(y, K, A[stage], h, out32=y32)
            fun(t + C[stage] * h, y32, K[stage])
        ops.rk_combine(y, K, B, h, out=ynew, out32=y32)
        fun(t + h, y32, K[6])
        nfev += 6
        y, ynew = ynew, y
        K[0].copy_(K[6])
    return OdeResult(y, float(t_grid[-1]), nfev, len(t_grid) - 1, 0)


def solve_rk45(fun, y0, t_span=(0.0, 1.0), rtol=1e-5, atol=1e-5, max_steps=1_000_000):
    """fun(t: float, y32: fp32 [n] device tensor, out: fp32 [n] view) writes dy/dt into `out`.
    y0: float64 [n] device tensor.  Returns OdeResult with y = float64 state at t_span[1]."""
    t, t_bound = float(t_span[0]), float(t_span[1])
    direction = 1.0 if t_bound >= t else -1.0
    eps = np.finfo(float).eps
    rtol = max(float(rtol), 100 * eps)      # solve_ivp's validate_tol
    n = y0.numel()
    dev = y0.device
    y = y0.detach().clone().contiguous()
    ynew = torch.empty_like(y)
    y32 = torch.empty(n, device=dev, dtype=torch.float32)
    K = torch.empty((7, n), device=dev, dtype=torch.float32)
    ws = ops.rk_workspace(dev)
    out = torch.empty(3, device=dev, dtype=torch.float64)

    ops.rk_combine(y, K, (), 0.0, out32=y32)
    fun(t, y32, K[0])
    nfev = 1
    # ---- select_initial_step
    interval = abs(t_bound - t)
    if interval == 0.0:
        return OdeResult(y, t, nfev, 0, 0)
    ops.rk_init_norms(y, K[0], None, rtol, atol, ws, out)
    s = out.tolist()
    d0, d1 = math.sqrt(s[0] / n), math.sqrt(s[1] / n)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    h0 = min(h0, interval)
    ops.rk_combine(y, K, (1.0,), h0 * direction, out32=y32)
    fun(t + h0 * direction, y32, K[1])
    nfev += 1
    ops.rk_init_norms(y, K[0], K[1], rtol, atol, ws, out)
    d2 = math.sqrt(out.tolist()[2] / n) / h0
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = max(1e-6, h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1, d2)) ** (1.0 / (ORDER + 1))
    h_abs = min(100 * h0, h1, interval)

    steps = rejected = 0
    while direction * (t - t_bound) < 0:
        if steps >= max_steps:
            raise RuntimeError("solve_rk45: max_steps exceeded")
        min_step = 10 * abs(np.nextafter(t, direction * np.inf) - t)
        if h_abs < min_step:
            h_abs = min_step
        step_rejected = False
        while True:
            if h_abs < min_step:
                raise RuntimeError("solve_rk45: required step size is less than spacing between numbers")
            h = h_abs * direction
            t_new = t + h
            if direction * (t_new - t_bound) > 0:
                t_new = t_bound
            h = t_new - t
            h_abs = abs(h)
            for stage in range(1, 6):
                ops.rk_combine(y, K, A[stage], h, out32=y32)
                fun(t + C[stage] * h, y32, K[stage])
            ops.rk_combine(y, K, B, h, out=ynew, out32=y32)
            fun(t + h, y32, K[6])
            nfev += 6
            ops.rk_error_norm(y, ynew, K, E, h, rtol, atol, ws, out)
            err = math.sqrt(out[0].item() / n)
            if not math.isfinite(err):
                raise FloatingPointError("solve_rk45: non-finite error estimate")
            if err < 1.0:
                factor = MAX_FACTOR if err == 0.0 else min(MAX_FACTOR, SAFETY * err ** EXPONENT)
                if step_rejected:
                    factor = min(1.0, factor)
                h_abs *= factor
                break
            h_abs *= max(MIN_FACTOR, SAFETY * err ** EXPONENT)
            step_rejected = True
            rejected += 1
        t = t_new
        y, ynew = ynew, y
        K[0].copy_(K[6])                    # first-same-as-last
        steps += 1
    return OdeResult(y, t, nfev, steps, rejected)

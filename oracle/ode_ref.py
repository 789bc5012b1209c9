"""Oracle restatement of torchdiffeq's ``odeint`` (TEST INFRASTRUCTURE ONLY).

**parity unpinned**: torchdiffeq is an un-vendored, un-versioned pip dependency of the
reference (requirements.txt:3) that is not installed in this image and cannot be fetched.
This file restates its published algorithm (fixed-grid euler/midpoint/rk4 and adaptive
dopri5) for the way the reference calls it:

    /root/reference/test_flow_latent.py:42-76
        odeint_adjoint(denoiser, x_0, t=tensor([1., 0.]), method=args.method,
                       atol=args.atol, rtol=args.rtol,
                       options={"step_size": h, "perturb": False} | {"dtype": float64})

Under ``torch.no_grad`` the adjoint wrapper is a plain ``odeint``.  Anchors (``tests/test_ode_ref.py``): the
Dormand-Prince tableau, single forced steps and the dense output against ``scipy.integrate.RK45`` (an independent
implementation of the same pair), the Runge-Kutta order conditions of both the 5th-order and the embedded 4th-order
weights, adaptive solves against ``solve_ivp``, analytic known answers (y'=-y, harmonic oscillator, orders, NFE counts).
"""
import torch


# ----------------------------------------------------------------------------- helpers
def _reverse(func):
    """Decreasing ``t`` => torchdiffeq integrates s=-t with f'(s,y) = -f(-s,y)."""
    return lambda s, y: -func(-s, y)


def _cast_time(func):
    """_PerturbFunc: the time handed to the user function is cast to y's dtype."""
    return lambda t, y: func(t.to(y.dtype), y)


def fixed_grid(t, step_size):
    """Grid of the fixed-step solvers: arange(n)*h + t0 with the LAST point forced to t_end."""
    start, end = t[0], t[-1]
    niters = torch.ceil((end - start) / step_size + 1).item()
    grid = torch.arange(0, niters, dtype=t.dtype) * step_size + start
    grid[-1] = t[-1]
    return grid


def _linear_interp(t0, t1, y0, y1, t):
    if t == t0:
        return y0
    if t == t1:
        return y1
    slope = (t - t0) / (t1 - t0)
    return y0 + slope * (y1 - y0)


# ----------------------------------------------------------------------------- fixed step
def _euler(f, t0, dt, t1, y0):
    return dt * f(t0, y0)


def _midpoint(f, t0, dt, t1, y0):
    half = 0.5 * dt
    return dt * f(t0 + half, y0 + f(t0, y0) * half)


def _rk4_38(f, t0, dt, t1, y0):
    """torchdiffeq's rk4 is the 3/8-rule variant."""
    k1 = f(t0, y0)
    k2 = f(t0 + dt / 3, y0 + dt * k1 / 3)
    k3 = f(t0 + dt * 2 / 3, y0 + dt * (k2 - k1 / 3))
    k4 = f(t1, y0 + dt * (k1 - k2 + k3))
    return (k1 + 3 * (k2 + k3) + k4) * dt * 0.125


_FIXED = {"euler": _euler, "midpoint": _midpoint, "rk4": _rk4_38}


def _integrate_fixed(step, f, y0, t, step_size):
    grid = fixed_grid(t, step_size)
    sol = [y0]
    j = 1
    for t0, t1 in zip(grid[:-1], grid[1:]):
        dt = t1 - t0
        y1 = y0 + step(f, t0, dt, t1, y0)
        while j < len(t) and t1 >= t[j]:
            sol.append(_linear_interp(t0, t1, y0, y1, t[j]))
            j += 1
        y0 = y1
    return torch.stack(sol, 0)


# ----------------------------------------------------------------------------- dopri5
_A = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
_B = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
_C_SOL = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0]
_C_ERR = [
    35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
    -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1.0 / 60.0,
]
_C_MID = [
    6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
    187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2,
]


def _rms(x):
    return x.pow(2).mean().sqrt()


def _lincomb(k, coefs, dt):
    out = None
    for ki, c in zip(k, coefs):
        if c == 0:
            continue
        term = ki * (c * dt)
        out = term if out is None else out + term
    return out


def _rk_step(f, y0, f0, t0, dt, t1):
    """One Dormand-Prince step; times are cast to y's dtype inside the step (as torchdiffeq does)."""
    t0, dt, t1 = t0.to(y0.dtype), dt.to(y0.dtype), t1.to(y0.dtype)
    k = [f0]
    yi = y0
    for a, b in zip(_A, _B):
        ti = t1 if a == 1.0 else t0 + a * dt
        yi = y0 + _lincomb(k, b, dt)
        k.append(f(ti, yi))
    y1 = yi  # FSAL: c_sol == last beta row
    err = _lincomb(k, _C_ERR, dt)
    return y1, k[-1], err, k


def _select_initial_step(f, t0, y0, order, rtol, atol, f0):
    t_dtype = t0.dtype
    t0 = t0.to(y0.dtype)
    scale = atol + y0.abs() * rtol
    d0, d1 = _rms(y0 / scale).abs(), _rms(f0 / scale).abs()
    h0 = torch.tensor(1e-6, dtype=y0.dtype) if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    h0 = h0.abs()
    f1 = f(t0 + h0, y0 + h0 * f0)
    d2 = (_rms((f1 - f0) / scale) / h0).abs()
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = torch.max(torch.tensor(1e-6, dtype=y0.dtype), h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1, d2)) ** (1.0 / float(order + 1))
    return torch.min(100 * h0, h1.abs()).to(t_dtype)


def _optimal_step(dt, ratio, safety=0.9, ifactor=10.0, dfactor=0.2, order=5):
    if ratio == 0:
        return dt * ifactor
    if ratio < 1:
        dfactor = 1.0
    ratio = ratio.to(dt.dtype)
    factor = min(ifactor, max(float(safety / ratio ** (1.0 / order)), dfactor))
    return dt * factor


def _interp_fit(y0, y1, k, dt):
    dt = dt.to(y0.dtype)
    y_mid = y0 + _lincomb(k, _C_MID, dt)
    f0, f1 = k[0], k[-1]
    a = 2 * dt * (f1 - f0) - 8 * (y1 + y0) + 16 * y_mid
    b = dt * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * y_mid
    c = dt * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * y_mid
    return [y0, dt * f0, c, b, a]


def _interp_eval(coef, t0, t1, t):
    x = ((t - t0) / (t1 - t0)).to(coef[0].dtype)
    total = coef[0] + x * coef[1]
    xp = x
    for c in coef[2:]:
        xp = xp * x
        total = total + xp * c
    return total


def _integrate_dopri5(f, y0, t, rtol, atol, stats):
    t = t.to(torch.float64)
    f0 = f(t[0], y0)
    dt = _select_initial_step(f, t[0], y0, 4, rtol, atol, f0)
    t0 = t1 = t[0]
    coef = [y0] * 5
    sol = [y0]
    for j in range(1, len(t)):
        while t[j] > t1:
            t_try = t1 + dt
            y_new, f_new, err, k = _rk_step(f, y0, f0, t1, dt, t_try)
            tol = atol + rtol * torch.max(y0.abs(), y_new.abs())
            ratio = _rms(err / tol).abs()
            stats["steps"] += 1
            if ratio <= 1:
                coef = _interp_fit(y0, y_new, k, dt)
                t0, t1, y0, f0 = t1, t_try, y_new, f_new
                stats["accepted"] += 1
            dt = _optimal_step(dt, ratio)
        sol.append(_interp_eval(coef, t0, t1, t[j]))
    return torch.stack(sol, 0)


# ----------------------------------------------------------------------------- entry point
@torch.no_grad()
def odeint(func, y0, t, method="dopri5", rtol=1e-5, atol=1e-5, options=None, stats=None):
    """``torchdiffeq.odeint`` as the reference uses it.  Returns ``[len(t), *y0.shape]``."""
    options = dict(options or {})
    stats = stats if stats is not None else {}
    stats.setdefault("steps", 0)
    stats.setdefault("accepted", 0)
    t = t.clone()
    f = func
    if len(t) > 1 and bool(t[0] > t[1]):
        t = -t
        f = _reverse(f)
    f = _cast_time(f)
    if method in _FIXED:
        return _integrate_fixed(_FIXED[method], f, y0, t, options.get("step_size"))
    if method == "dopri5":
        return _integrate_dopri5(f, y0, t, rtol, atol, stats)
    raise NotImplementedError(f"oracle restates euler/midpoint/rk4/dopri5 only, not {method!r}")

"""ODE solvers of the sampling loop (what the reference gets from torchdiffeq at test_flow_latent.py:23,61-73).

``odeint(func, y0, t, method=, rtol=, atol=, options=)`` follows torchdiffeq's contract for the call the reference
makes -- ``t = tensor([1., 0.])`` (decreasing => time is negated internally), fixed-grid ``euler / midpoint / rk4``
with ``options={"step_size": h}`` and adaptive ``dopri5`` with fp64 time -- and returns ``[len(t), *y0.shape]``.
It is host control flow over device tensors and works for any callable.

The production path is ``GraphedFixedGrid``: for the HIP DiT the whole solver interval (time-grid advance, velocity
field incl. CFG doubling, and the update x <- x + dt*v fused into the model's last kernel) is captured once in a
hipGraph and replayed per interval; the time grid lives in device memory so the replay needs no host value.
"""
import math

import os

import torch

from . import hip


class _SolverCache(dict):
    """{(shape, cfg, ...): GraphedFixedGrid} held by the model object (so it dies with the model) -- but captured graphs must never be
    copied or serialised with it: copy.deepcopy(model) / torch.save(model) see an EMPTY cache."""

    def __deepcopy__(self, memo):
        return _SolverCache()

    def __reduce__(self):
        return (_SolverCache, ())

ADAPTIVE_SOLVER = ["dopri5", "dopri8", "adaptive_heun", "bosh3"]
FIXER_SOLVER = ["euler", "rk4", "midpoint", "stochastic"]


# ============================================================================ generic odeint
def fixed_grid(t, step_size):
    """torchdiffeq grid constructor: arange(n)*h + t0 with the last point forced to t_end (t already increasing)."""
    niters = torch.ceil((t[-1] - t[0]) / step_size + 1).item()
    grid = torch.arange(0, niters, dtype=t.dtype, device=t.device) * step_size + t[0]
    grid[-1] = t[-1]
    return grid


def _euler_step(f, t0, dt, t1, y0):
    return dt * f(t0, y0)


def _midpoint_step(f, t0, dt, t1, y0):
    half = 0.5 * dt
    return dt * f(t0 + half, y0 + f(t0, y0) * half)


def _rk4_step(f, t0, dt, t1, y0):  # torchdiffeq's "rk4" is the 3/8 rule
    k1 = f(t0, y0)
    k2 = f(t0 + dt / 3, y0 + dt * k1 / 3)
    k3 = f(t0 + dt * 2 / 3, y0 + dt * (k2 - k1 / 3))
    k4 = f(t1, y0 + dt * (k1 - k2 + k3))
    return (k1 + 3 * (k2 + k3) + k4) * dt * 0.125


_STEP = {"euler": _euler_step, "midpoint": _midpoint_step, "rk4": _rk4_step}

# Dormand-Prince 5(4), Shampine's dense-output midpoint coefficients
_DP_A = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
_DP_B = ((1 / 5,), (3 / 40, 9 / 40), (44 / 45, -56 / 15, 32 / 9), (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
         (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656), (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84))
_DP_E = (35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 + 12231 / 42400,
         11 / 84 - 649 / 6300, -1.0 / 60.0)
_DP_MID = (6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
           187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2)


def _rms(x):
    return x.pow(2).mean().sqrt()


def _comb(ks, coefs, dt):
    acc = None
    for k, c in zip(ks, coefs):
        if c != 0:
            term = k * (c * dt)
            acc = term if acc is None else acc + term
    return acc


# Bogacki-Shampine 3(2) ("bosh3") and Heun-Euler 2(1) ("adaptive_heun") as torchdiffeq tabulates them; the midpoint weights feed the
# same quartic dense-output fit as dopri5.  bosh3's pair is scipy's RK23 (tests/test_ode_ref.py checks the tableau against it).
_BS_A = (1 / 2, 3 / 4, 1.0)
_BS_B = ((1 / 2,), (0.0, 3 / 4), (2 / 9, 1 / 3, 4 / 9))
_BS_E = (2 / 9 - 7 / 24, 1 / 3 - 1 / 4, 4 / 9 - 1 / 3, -1 / 8)
_BS_MID = (0.0, 0.5, 0.0, 0.0)
_AH_A = (1.0,)
_AH_B = ((1.0,),)
_AH_SOL = (0.5, 0.5)
_AH_E = (0.5, -0.5)
_AH_MID = (0.5, 0.0)


# Prince-Dormand 8(7) with 13 stages ("dopri8", RK8(7)13M), the pair torchdiffeq tabulates: 12 stage rows, then the 8th-order weights as a
# 13th row so that the last stage is the solution (its derivative is f1 of the next step).  The rational coefficients are the published
# ones (Prince & Dormand 1981); tests/test_ode_ref.py checks them against ALL 200 rooted-tree order conditions through order 8 (worst
# residual 1.1e-15), the embedded weights against all 85 through order 7, and that neither goes one order further.  torchdiffeq's own
# midpoint weights for the dense output could not be restated offline (parity unpinned); _D8_MID is OUR midpoint: the minimum-norm weights
# on stages {0, 5..11} that meet the five quadrature conditions at theta = 1/2 -- which, by the tableau's simplifying assumptions, satisfy
# all 17 order conditions through order 5 (tested) -- one order more than the quartic fit that consumes it can use.
_D8_A = (1 / 18, 1 / 12, 1 / 8, 5 / 16, 3 / 8, 59 / 400, 93 / 200, 5490023248 / 9719169821, 13 / 20, 1201146811 / 1299019798, 1.0, 1.0, 1.0)
_D8_SOL = (14005451 / 335480064, 0.0, 0.0, 0.0, 0.0, -59238493 / 1068277825, 181606767 / 758867731, 561292985 / 797845732,
           -1041891430 / 1371343529, 760417239 / 1151165299, 118820643 / 751138087, -528747749 / 2220607170, 1 / 4)
_D8_EMB = (13451932 / 455176623, 0.0, 0.0, 0.0, 0.0, -808719846 / 976000145, 1757004468 / 5645159321, 656045339 / 265891186,
           -3867574721 / 1518517206, 465885868 / 322736535, 53011238 / 667516719, 2 / 45, 0.0)
_D8_B = (
    (1 / 18,),
    (1 / 48, 1 / 16),
    (1 / 32, 0.0, 3 / 32),
    (5 / 16, 0.0, -75 / 64, 75 / 64),
    (3 / 80, 0.0, 0.0, 3 / 16, 3 / 20),
    (29443841 / 614563906, 0.0, 0.0, 77736538 / 692538347, -28693883 / 1125000000, 23124283 / 1800000000),
    (16016141 / 946692911, 0.0, 0.0, 61564180 / 158732637, 22789713 / 633445777, 545815736 / 2771057229, -180193667 / 1043307555),
    (39632708 / 573591083, 0.0, 0.0, -433636366 / 683701615, -421739975 / 2616292301, 100302831 / 723423059, 790204164 / 839813087,
     800635310 / 3783071287),
    (246121993 / 1340847787, 0.0, 0.0, -37695042795 / 15268766246, -309121744 / 1061227803, -12992083 / 490766935, 6005943493 / 2108947869,
     393006217 / 1396673457, 123872331 / 1001029789),
    (-1028468189 / 846180014, 0.0, 0.0, 8478235783 / 508512852, 1311729495 / 1432422823, -10304129995 / 1701304382, -48777925059 / 3047939560,
     15336726248 / 1032824649, -45442868181 / 3398467696, 3065993473 / 597172653),
    (185892177 / 718116043, 0.0, 0.0, -3185094517 / 667107341, -477755414 / 1098053517, -703635378 / 230739211, 5731566787 / 1027545527,
     5232866602 / 850066563, -4093664535 / 808688257, 3962137247 / 1805957418, 65686358 / 487910083),
    (403863854 / 491063109, 0.0, 0.0, -5068492393 / 434740067, -411421997 / 543043805, 652783627 / 914296604, 11173962825 / 925320556,
     -13158990841 / 6184727034, 3936647629 / 1978049680, -160528059 / 685178525, 248638103 / 1413531060, 0.0),
    _D8_SOL,
)
_D8_E = tuple(a - b for a, b in zip(_D8_SOL, _D8_EMB)) + (0.0,)
_D8_MID = (0.04075652697815933, 0.0, 0.0, 0.0, 0.0, 0.14575312352778258, 0.2349319583967853, 0.07730057792519462, 0.015741038709153957,
           -0.015256809590672644, 3.823380238289129e-05, 0.0007353502512138821, 0.0, 0.0)


class Dopri5:
    """Adaptive Dormand-Prince as torchdiffeq runs it: fp64 time, RMS error norm over the WHOLE state tensor (step sizes
    are batch-coupled), accept iff ratio <= 1, dt *= min(10, max(0.9 ratio^-1/order, 0.2|1)), FSAL, steps not clipped to the end
    time (the model is queried slightly past it), 4th-order dense output evaluated at the requested time.

    The same adaptive Runge-Kutta driver runs the other embedded pairs of torchdiffeq's RKAdaptiveStepsizeODESolver through the class
    attributes below (``Bosh3``, ``AdaptiveHeun``): stage nodes A, stage rows B, solution weights SOL (None = last row of B, FSAL),
    error weights E, midpoint weights MID, ORDER (exponent of the step controller and of the initial-step heuristic)."""

    A, B, SOL, E, MID, ORDER = _DP_A, _DP_B, None, _DP_E, _DP_MID, 5

    def __init__(self, f, y0, t0, rtol, atol, max_num_steps=2 ** 31 - 1):
        self.f, self.rtol, self.atol, self.max_num_steps = f, rtol, atol, max_num_steps
        self.nfe_steps = 0
        self.accepted = 0
        # device fast path: fp32 state on the GPU and at most 8 stage derivatives per combination -> every stage sum is ONE
        # lfm_lincomb launch, the error ratio ONE fused reduction (lfm_rk_error_norm) whose 4-byte result is the step's only device -> host read;
        # time is kept twice, as device tensors for the kernels and as host doubles for the control flow (same IEEE arithmetic, no read-back)
        self._dev = bool(y0.is_cuda and y0.dtype == torch.float32 and y0.is_contiguous() and y0.numel() % 4 == 0 and len(self.E) <= 8
                         and float(rtol) == rtol and float(atol) == atol)
        self._y_like = y0
        if self._dev:
            dv = y0.device
            self._cB = [torch.tensor(b, dtype=torch.float32, device=dv) for b in self.B]
            self._cE = torch.tensor(self.E, dtype=torch.float32, device=dv)
            self._cSOL = None if self.SOL is None else torch.tensor(self.SOL, dtype=torch.float32, device=dv)
            self._scratch = torch.empty(1024, dtype=torch.float32, device=dv)
            self._ratio = torch.empty(1, dtype=torch.float32, device=dv)
        f0 = self._stage(f(t0, y0))
        self.y0, self.f0 = y0, f0
        self.t0 = self.t1 = t0
        self._t0_h = self._t1_h = float(t0)
        self.dt = self._initial_step(t0, y0, f0, order=self.ORDER - 1)  # (property: also sets the host copy)
        self._last = None
        self.coef = [y0] * 5

    @property
    def dt(self):
        return self._dt

    @dt.setter
    def dt(self, value):  # set from outside (initial step, tests forcing a step): one read-back keeps the host copy exact
        self._dt, self._dt_h = value, float(value)

    def _initial_step(self, t0, y0, f0, order=4):
        scale = self.atol + y0.abs() * self.rtol
        d0, d1 = _rms(y0 / scale), _rms(f0 / scale)
        h0 = torch.full_like(d0, 1e-6) if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        f1 = self.f(t0.to(y0.dtype) + h0, y0 + h0 * f0)
        d2 = _rms((f1 - f0) / scale) / h0
        if d1 <= 1e-15 and d2 <= 1e-15:
            h1 = torch.max(torch.full_like(h0, 1e-6), h0 * 1e-3)
        else:
            h1 = (0.01 / torch.max(d1, d2)) ** (1.0 / (order + 1))
        return torch.min(100 * h0, h1).to(t0.dtype)

    def _stage(self, k):
        """A stage derivative as the device kernels need it.  odeint() is a generic torchdiffeq-style API: `func` may return an expanded / permuted view,
        another dtype (the reverse-time wrapper preserves strides) or a tensor on another device.  lfm_lincomb / lfm_rk_error_norm take raw pointers to dense
        fp32 arrays of y0's length, so anything else is coerced here (one copy), and a result that cannot be coerced turns the device path off."""
        if not self._dev:
            return k
        y0 = self.y0 if hasattr(self, "y0") else None
        ref = y0 if y0 is not None else self._y_like
        if torch.is_tensor(k) and k.shape == ref.shape and k.device == ref.device:
            if k.dtype != ref.dtype or not k.is_contiguous():
                k = k.to(ref.dtype).contiguous()
            return k
        self._dev = False  # a broadcastable / off-device result: the eager combinations below handle it as torch would
        return k

    def _lin(self, base, ks, coef_dev, coef_host, dty):
        """base + dty * sum coef_j k_j: one lincomb launch on the device path, eager torch otherwise."""
        if self._dev and len(ks) <= 8:
            out = torch.empty_like(base)
            hip.lincomb(out, base, ks, coef_dev[: len(ks)], dty)
            return out
        return base + _comb(ks, coef_host, dty)

    def _step(self):
        y0, f0, t0, dt = self.y0, self.f0, self.t1, self.dt
        t1 = t0 + dt
        t0y, dty, t1y = t0.to(y0.dtype), dt.to(y0.dtype), t1.to(y0.dtype)
        k = [f0]
        yi = y0
        for i, (a, b) in enumerate(zip(self.A, self.B)):
            yi = self._lin(y0, k, self._cB[i] if self._dev else None, b, dty)
            k.append(self._stage(self.f(t1y if a == 1.0 else t0y + a * dty, yi)))
        # FSAL pairs: the last stage IS the solution; otherwise (adaptive_heun) the solution is its own combination and -- as in
        # torchdiffeq's _runge_kutta_step -- the last stage's derivative still serves as f1 of the next step
        y1 = yi if self.SOL is None else self._lin(y0, k, self._cSOL if self._dev else None, self.SOL, dty)
        f1 = k[-1]
        if self._dev:
            hip.rk_error_norm(y0, y1, k, self._cE, dty, self.rtol, self.atol, self._scratch, self._ratio)
            ratio_h = float(self._ratio)  # the one device->host read of the step
        else:
            err = _comb(k, self.E, dty)
            tol = self.atol + self.rtol * torch.max(y0.abs(), y1.abs())
            ratio_h = float(_rms(err / tol))
        self.nfe_steps += 1
        if ratio_h <= 1:
            # the quartic dense output is only ever evaluated for the step that contains a requested time: keep what it needs, fit lazily
            self._last = (y0, y1, f0, f1, k, dty)
            self.coef = None
            self.t0, self.t1, self.y0, self.f0 = t0, t1, y1, f1
            self._t0_h, self._t1_h = self._t1_h, self._t1_h + self._dt_h
            self.accepted += 1
        if ratio_h == 0:
            factor = 10.0
        else:
            factor = min(10.0, max(0.9 / ratio_h ** (1.0 / self.ORDER), 1.0 if ratio_h < 1 else 0.2))
        self._dt, self._dt_h = dt * factor, self._dt_h * factor  # both copies, no read-back

    def _fit(self):
        if self.coef is None:
            y0, y1, f0, f1, k, dty = self._last
            y_mid = y0 + _comb(k, self.MID, dty)
            a = 2 * dty * (f1 - f0) - 8 * (y1 + y0) + 16 * y_mid
            b = dty * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * y_mid
            c = dty * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * y_mid
            self.coef = [y0, dty * f0, c, b, a]
        return self.coef

    def advance(self, t_next):
        n = 0
        t_next_h = float(t_next)
        while t_next_h > self._t1_h:
            assert n < self.max_num_steps
            self._step()
            n += 1
        coef = self._fit()
        x = ((t_next - self.t0) / (self.t1 - self.t0)).to(coef[0].dtype)
        total = coef[0] + x * coef[1]
        xp = x
        for c in coef[2:]:
            xp = xp * x
            total = total + xp * c
        return total


class Bosh3(Dopri5):
    A, B, SOL, E, MID, ORDER = _BS_A, _BS_B, None, _BS_E, _BS_MID, 3


class AdaptiveHeun(Dopri5):
    A, B, SOL, E, MID, ORDER = _AH_A, _AH_B, _AH_SOL, _AH_E, _AH_MID, 2


class Dopri8(Dopri5):
    """13 evaluations per step; the time a result is asked for is reached by the same quartic dense output as the other pairs, through
    _D8_MID (ours, see above) -- the one place where this solver can differ from torchdiffeq's dopri8, at the level of the interpolation error."""
    A, B, SOL, E, MID, ORDER = _D8_A, _D8_B, None, _D8_E, _D8_MID, 8


_ADAPTIVE = {"dopri5": Dopri5, "dopri8": Dopri8, "bosh3": Bosh3, "adaptive_heun": AdaptiveHeun}


def _perturbed(f):
    """torchdiffeq's _PerturbFunc for the fixed-grid solvers under options['perturb']: the evaluation at the START of a step is taken one
    ulp after it, the one at the END one ulp before it (so a field that is discontinuous at grid points is sampled inside the step)."""

    def g(t, y, perturb=0):
        t = t.to(y.dtype)  # torchdiffeq casts the time to the state's dtype FIRST: an ulp of a float64 grid would be rounded away in a float32 state
        if perturb > 0:
            t = torch.nextafter(t, t + 1)
        elif perturb < 0:
            t = torch.nextafter(t, t - 1)
        return f(t, y)

    return g


def _euler_step_p(f, t0, dt, t1, y0):
    return dt * f(t0, y0, perturb=1)


def _midpoint_step_p(f, t0, dt, t1, y0):
    half = 0.5 * dt
    return dt * f(t0 + half, y0 + f(t0, y0, perturb=1) * half)


def _rk4_step_p(f, t0, dt, t1, y0):
    k1 = f(t0, y0, perturb=1)
    k2 = f(t0 + dt / 3, y0 + dt * k1 / 3)
    k3 = f(t0 + dt * 2 / 3, y0 + dt * (k2 - k1 / 3))
    k4 = f(t1, y0 + dt * (k1 - k2 + k3), perturb=-1)
    return (k1 + 3 * (k2 + k3) + k4) * dt * 0.125


_STEP_P = {"euler": _euler_step_p, "midpoint": _midpoint_step_p, "rk4": _rk4_step_p}


@torch.no_grad()
def odeint(func, y0, t, *, rtol=1e-7, atol=1e-9, method=None, options=None, stats=None):
    method = method or "dopri5"
    options = dict(options or {})
    t = t.detach().clone()
    f = func
    if len(t) > 1 and bool(t[0] > t[1]):  # decreasing time: integrate s = -t with f'(s, y) = -f(-s, y)
        t = -t
        f = lambda s, y, _f=func: -_f(-s, y)  # noqa: E731
    g = lambda s, y, _f=f: _f(s.to(y.dtype), y)  # noqa: E731  -- the model always sees time in y's dtype
    if method in _STEP:
        if "step_size" not in options:
            raise ValueError("fixed-grid solvers need options['step_size']")
        grid = fixed_grid(t, options["step_size"])
        step = _STEP[method]
        if options.get("perturb"):
            g, step = _perturbed(g), _STEP_P[method]
        sol, j = [y0], 1
        tl = t.tolist()
        gl = grid.tolist()
        for i in range(len(gl) - 1):
            t0, t1 = grid[i], grid[i + 1]
            y1 = y0 + step(g, t0, t1 - t0, t1, y0)
            while j < len(tl) and gl[i + 1] >= tl[j]:
                if tl[j] == gl[i + 1]:
                    sol.append(y1)
                elif tl[j] == gl[i]:
                    sol.append(y0)
                else:
                    sol.append(y0 + (t[j] - t0) / (t1 - t0) * (y1 - y0))
                j += 1
            y0 = y1
        return torch.stack(sol, 0)
    if method in _ADAPTIVE:
        t = t.to(torch.float64)
        solver = _ADAPTIVE[method](g, y0, t[0], rtol, atol)
        sol = [y0] + [solver.advance(t[j]) for j in range(1, len(t))]
        if stats is not None:
            stats.update(steps=solver.nfe_steps, accepted=solver.accepted, nfe=2 + len(solver.A) * solver.nfe_steps)
        return torch.stack(sol, 0)
    raise NotImplementedError(f"method {method!r}: euler / midpoint / rk4 / dopri5 / dopri8 / bosh3 / adaptive_heun are built")


def torchdiffeq_euler_grid(step_size, device=None):
    """Model-time grid of ``odeint(..., t=[1,0], method='euler', options={'step_size': h})``: returns (ts[n+1], dts[n]) with
    t_k = -(k*h - 1) evaluated in fp32 exactly like the grid constructor and x_{k+1} = x_k + dts[k] * v(ts[k], x_k)."""
    s = fixed_grid(torch.tensor([-1.0, -0.0]), step_size)
    ts = -s
    dts = -(s[1:] - s[:-1])
    return ts.to(device), dts.to(device)


# ============================================================================ fused, graph-captured fixed grid
def fused_fixed_grid_available(model, x):
    from .models.DiT import DiT
    from .models.EDM import DhariwalUNet
    from .models.unet import UNetModel

    inner = getattr(model, "model", None) if type(model).__name__ == "WrapperCondFlow" else model  # downstream-task conditioning wrapper
    return isinstance(inner, (DiT, UNetModel, DhariwalUNet)) and x.is_cuda and not inner.training


class GraphedFixedGrid:
    """One captured hipGraph per (model, batch, cfg) for the intervals of a fixed time grid.

    euler interval :  advance(t, dt) -> DiT forward with  x <- x + dt * v(t, x)  fused into its final kernel.
    heun interval  :  advance -> d1 = v(t, x);  xp = x + dt*d1;  d2 = v(t_next, xp);  x <- x + dt*(0.5 d1 + 0.5 d2).
    """

    def __init__(self, model, batch, y=None, cfg_scale=1.0, use_cfg=False, graph=True, resolution=None):
        from .models.DiT import DiT

        self.is_dit = isinstance(model, DiT)
        dev = next(model.parameters()).device
        C = model.in_channels
        R = model.img_resolution if self.is_dit else (resolution or model.image_size)
        if use_cfg and not self.is_dit:
            raise NotImplementedError("classifier-free guidance needs forward_with_cfg, which the origin-ADM UNet does not have")
        self.model, self.batch, self.dev = model, batch, dev
        self.y = None if y is None else y.to(dev, torch.long).clone()  # own the buffer the captured graph reads (never alias the caller's)
        self.use_cfg, self.cfg_scale = bool(use_cfg), float(cfg_scale)
        self.x = torch.zeros(batch, C, R, R, device=dev)
        self.d1 = torch.zeros_like(self.x)
        self.d2 = torch.zeros_like(self.x)
        self.xp = torch.zeros_like(self.x)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.tcur = torch.zeros(1, device=dev)
        self.tnext = torch.zeros(1, device=dev)
        self.dt = torch.zeros(1, device=dev)
        self.c1 = torch.ones(1, device=dev)
        self.c2 = torch.full((2,), 0.5, device=dev)
        self.max_intervals = 4096
        self.ts = torch.zeros(self.max_intervals + 1, device=dev)  # persistent: the captured graphs read these addresses
        self.dts = torch.zeros(self.max_intervals, device=dev)
        self.n_intervals = 0
        self.use_graph = graph
        self.graphs = {}
        self._graph_gen = -1  # model buffer generation the graphs were captured against
        # per-grid conditioning table (DiT.cond_table): unconditional DiT at scalar time only.  The captured graphs hold its address, so it lives in a
        # persistent buffer that is only re-FILLED when the grid or the weights change (re-allocated, and the graphs dropped, when a longer grid arrives)
        self.use_cond_table = self.is_dit and self.y is None and not self.use_cfg and os.environ.get("LFM_COND_TABLE", "1") != "0"
        self.cond_buf = None
        self._cond_rows = 0  # rows of the table the evaluations may index (0: no table in use)
        self._cond_key = None
        self._grid_host = ()

    def set_grid(self, ts, dts):
        """Copy a time grid (n+1 times, n signed steps) INTO the persistent device buffers the captured graphs point at."""
        n = dts.numel()
        if ts.numel() != n + 1 or n < 1 or n > self.max_intervals:
            raise ValueError(f"grid with {ts.numel()} times / {n} steps (max {self.max_intervals} intervals)")
        self.ts[: n + 1].copy_(ts.to(torch.float32))
        self.dts[:n].copy_(dts.to(torch.float32))
        self.n_intervals = n
        if self.use_cond_table:  # the table's cache key; only the table needs the grid on the host (a device-resident Karras grid is read back once per solve)
            self._grid_host = tuple(float(v) for v in ts.detach().cpu().tolist())
            self._refresh_cond()

    def _refresh_cond(self):
        """(Re)fill the conditioning table when the grid or the model's weights changed since it was written (run() asks too: a load_state_dict between
        two solves on the same grid must not leave stale rows behind)."""
        if not self.use_cond_table or self.n_intervals < 1:
            return
        n = self.n_intervals
        if self._cond_key == (self._grid_host, getattr(self.model, "_gen", 0)):
            return
        import ctypes as _C
        shape = self.model.shape_struct()
        if hip.lib().lfm_dit_cond_table_bytes(_C.byref(shape), n + 1) > self.COND_TABLE_MAX_BYTES:  # a 1.6 % saving is not worth gigabytes
            if self._cond_rows:
                self.graphs.clear()  # captured with table rows: re-capture without
            self._cond_rows = 0
            self._cond_key = (self._grid_host, getattr(self.model, "_gen", 0))
            return
        table = self.model.cond_table(self.ts[: n + 1], self.batch)  # may (re)pack weights / size the workspace: read the generation after it
        if self.cond_buf is None or self.cond_buf.numel() < table.numel():
            self.cond_buf = table
            self.graphs.clear()  # they point at the old buffer
        else:
            self.cond_buf[: table.numel()].copy_(table)
        if self._cond_rows != n + 1:
            self.graphs.clear()  # the row count is baked into the captured calls (lfm_dit_call.cond_rows)
        self._cond_rows = n + 1
        self._cond_key = (self._grid_host, getattr(self.model, "_gen", 0))

    def _cond(self, offset):
        """(table, interval counter, row offset) of an evaluation at grid time ts[step + offset]; _advance has already incremented the counter."""
        return (self.cond_buf, self.step, offset, self._cond_rows) if self._cond_active() else None

    COND_TABLE_MAX_BYTES = 1 << 30  # ~2 MB per grid time for DiT-L/2: 50-500 intervals; finer grids (step_size 1e-3) recompute the conditioning per evaluation

    def _cond_active(self):
        return self.use_cond_table and self.cond_buf is not None and self._cond_rows > 0

    def _advance(self):
        hip.check(hip.lib().lfm_grid_advance(hip.ptr(self.ts), hip.ptr(self.dts), hip.ptr(self.step), hip.ptr(self.tcur), hip.ptr(self.tnext),
                                             hip.ptr(self.dt), hip.stream_ptr(self.dev)), "lfm_grid_advance")

    def _velocity(self, t, x, out=None, grid_offset=None):
        if self.is_dit:
            return self.model._run(t, x, self.y, self.use_cfg, self.cfg_scale, out=out, cond=None if grid_offset is None else self._cond(grid_offset))
        v = self.model(t, x, self.y)  # host-sequenced UNet: its launches are captured like any others
        if out is not None:
            out.copy_(v)
            return out
        return v

    def _euler(self):
        self._advance()
        if self.is_dit:  # x <- x + dt*v fused into the model's last kernel
            self.model._run(self.tcur, self.x, self.y, self.use_cfg, self.cfg_scale, out=self.x, axpy_base=self.x, axpy_dt=self.dt, cond=self._cond(-1))
        else:
            hip.lincomb(self.x, self.x, [self._velocity(self.tcur, self.x)], self.c1, self.dt)

    def _heun(self):
        self._advance()
        d1 = self._velocity(self.tcur, self.x, out=self.d1, grid_offset=-1)  # at ts[k] (the counter already reads k + 1)
        hip.lincomb(self.xp, self.x, [d1], self.c1, self.dt)
        d2 = self._velocity(self.tnext, self.xp, out=self.d2, grid_offset=0)  # at ts[k + 1]
        hip.lincomb(self.x, self.x, [d1, d2], self.c2, self.dt)

    def _get(self, kind):
        fn = self._euler if kind == "euler" else self._heun
        if not self.use_graph:
            return fn
        if self._graph_gen != getattr(self.model, "_gen", 0):
            # weights were re-packed / a workspace was re-allocated (load_state_dict, .to(), bigger batch elsewhere): the
            # captured graphs hold stale device pointers -> drop them
            self.graphs.clear()
        if kind not in self.graphs:
            # warm up on a side stream (packs weights, sizes the workspace, sets kernel attributes), then capture
            side = torch.cuda.Stream(self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                saved = (self.x.clone(), self.step.clone())
                fn()
                self.x.copy_(saved[0])
                self.step.copy_(saved[1])
            torch.cuda.current_stream(self.dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            # thread_local: only THIS thread's calls are policed during capture -- a multi-rank run has the RCCL watchdog thread polling
            # events of the previous batch's image all-gather (side stream), which the default global mode would turn into a capture error
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
            self.x.copy_(saved[0])
            self.step.copy_(saved[1])
            self.graphs[kind] = g
            self._graph_gen = getattr(self.model, "_gen", 0)
        return self.graphs[kind].replay

    @torch.no_grad()
    def run(self, x0, heun_limit=0):
        """Integrate over the whole grid.  heun_limit = number of leading intervals... see karras_sample: the corrector is
        applied on interval i iff i < heun_limit - 1;  heun_limit = 0 means plain Euler."""
        assert self.n_intervals > 0, "set_grid first"
        n = self.n_intervals
        self._refresh_cond()
        self.x.copy_(x0)
        self.step.zero_()
        n_heun = max(0, min(n, heun_limit - 1)) if heun_limit else 0
        self.last_plan = {"heun": n_heun, "euler": n - n_heun, "nfe": 2 * n_heun + (n - n_heun)}  # what this call replays
        if n_heun:
            h = self._get("heun")
            for _ in range(n_heun):
                h()
        if n - n_heun:
            e = self._get("euler")
            for _ in range(n - n_heun):
                e()
        return self.x


# The captured solvers of a model live ON the model object (not in a table keyed by id(model): a recycled id after GC would replay
# another model's graph), so they are collected with it.


def _label_rows(model):
    if hasattr(model, "y_embedder"):
        return model.y_embedder.get_in_channels()
    return getattr(model, "num_classes", None)


def _fused(model, x, model_kwargs):
    y = model_kwargs.get("y")
    cfg_scale = float(model_kwargs.get("cfg_scale", 1.0))
    use_cfg = cfg_scale > 1.0
    if y is not None and _label_rows(model):
        hip.check_labels(y, _label_rows(model), type(model).__name__)  # once, before the labels go into the captured graph's buffer
    per_model = model.__dict__.setdefault("_fused_solvers", _SolverCache())
    key = (tuple(x.shape), use_cfg, cfg_scale, y is not None, x.device)
    fg = per_model.get(key)
    if fg is None:
        if len(per_model) > 8:
            per_model.clear()
        fg = GraphedFixedGrid(model, x.shape[0], y=y, cfg_scale=cfg_scale, use_cfg=use_cfg, resolution=x.shape[-1])
        per_model[key] = fg
    elif y is not None:
        fg.y.copy_(y)  # labels are read by the captured kernels from this (solver-owned) buffer
    return fg


def concurrency_twin(module):
    """A second handle on the SAME weights with its own device scratch (workspace, captured solver graphs), so that two batches can be in flight on two
    HIP streams: a sampling job's batches are independent (test_flow_latent_ddp.py:116-146), and two chip-filling evaluations launched side by side run
    ~5 % faster than one after the other -- while one stream's workgroups sit in their HBM-bound GEMM epilogues, the other's are in their MFMA main loops
    (profiles/r04_two_batches_in_flight.txt).  Works for the DiT family, the VAE and the two UNets (modules whose scratch is `_ws` / `_scratch` / `_conv_ws` / `_fused_solvers`).  Take the twin AFTER
    the weights are final: it shares the packed weight buffers of the original and does not follow a later load_state_dict.  The stream a twin (or the
    original) is then driven on must first wait for the stream its setup ran on (packing, set_grid's conditioning tables): `s.wait_stream(current)`."""
    import copy

    if hasattr(module, "_pack") and getattr(module, "_packed", None) is None and any(p.is_cuda for p in module.parameters()):
        module._pack()  # pack once, share the buffers
    t = copy.copy(module)
    t.__dict__.pop("_fused_solvers", None)
    for scratch in ("_ws", "_scratch", "_conv_ws", "_film_all"):  # DiT / VAE workspace; the UNets' GroupNorm / split-K scratch and FiLM rows (allocated on first use)
        if scratch in t.__dict__:
            setattr(t, scratch, None)
    return t


def sample_fixed_grid_fused(model, x, sigmas, model_kwargs, heun_limit=0):
    """Karras-style grid: t_i = sigmas[i], dt_i = sigmas[i+1] - sigmas[i] (karras_sample.py:30,102-117,151-159)."""
    fg = _fused(model, x, model_kwargs)
    sig = sigmas.to(torch.float32)
    fg.set_grid(sig, sig[1:] - sig[:-1])
    return fg.run(x, heun_limit=heun_limit).clone()


def sample_torchdiffeq_euler_fused(model, x, step_size, model_kwargs):
    """The reference's production Euler: odeint(..., t=[1,0], method='euler', options={'step_size': h})."""
    fg = _fused(model, x, model_kwargs)
    ts, dts = torchdiffeq_euler_grid(step_size)
    fg.set_grid(ts, dts)
    return fg.run(x).clone()


_ = math

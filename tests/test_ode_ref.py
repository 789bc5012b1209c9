"""Independent anchors for the torchdiffeq restatement (oracle/ode_ref.py) and the product solver (lfm_amd/solvers.py).  CPU only.

torchdiffeq is not installable here, so ``dopri5`` stays **parity unpinned** against torchdiffeq itself.  What CAN be pinned, against
sources that share no code or author with either file:

* the Dormand-Prince tableau (nodes, stage matrix, 5th-order weights) against ``scipy.integrate.RK45`` (scipy 1.15, an independent
  implementation of the same published pair);
* the 4th-order dense output -- torchdiffeq fits a quartic through (y0, f0, y_mid, y1, f1) with Shampine's midpoint weights -- against
  scipy's dense-output polynomial ``RK45.P`` (also Shampine's), on a nonlinear field with a FORCED identical step;
* single steps with forced identical step sizes against scipy's own ``rk_step`` on a stiff-ish nonlinear field;
* the Runge-Kutta order conditions themselves: the 5th-order weights satisfy all 17 conditions through order 5, and the embedded
  weights ``c_sol - c_err`` (torchdiffeq's error estimator is NOT scipy's ``E``: it embeds a different 4th-order solution) satisfy
  the 8 conditions through order 4 -- so ``err`` really is an O(h^5) local error estimate;
* a full adaptive solve against ``scipy.integrate.solve_ivp(method="RK45")`` at the same tolerances (different controller and
  estimator, so the step sequences differ; both must sit within the tolerance band of a tight reference solution).
"""
import functools
import itertools

import numpy as np
import pytest
import torch
from scipy.integrate import RK45, solve_ivp
from scipy.integrate._ivp.rk import rk_step

from lfm_amd import solvers as prod
from oracle import ode_ref


def _tableaus():
    return {
        "oracle": dict(a=ode_ref._A, b=ode_ref._B, sol=ode_ref._C_SOL, err=ode_ref._C_ERR, mid=ode_ref._C_MID),
        "product": dict(a=list(prod._DP_A), b=[list(r) for r in prod._DP_B], sol=list(prod._DP_B[-1]) + [0.0], err=list(prod._DP_E),
                        mid=list(prod._DP_MID)),
    }


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_tableau_equals_scipy_rk45(which):
    tb = _tableaus()[which]
    np.testing.assert_allclose(RK45.C[1:], tb["a"][:5], rtol=0, atol=1e-16)
    assert tb["a"][5] == 1.0  # the FSAL stage sits at t1
    for i in range(1, 6):
        np.testing.assert_allclose(RK45.A[i, :i], tb["b"][i - 1], rtol=1e-15, atol=1e-16)
    np.testing.assert_allclose(RK45.B, tb["b"][5], rtol=1e-15, atol=1e-16)     # last stage row == 5th-order weights (FSAL)
    np.testing.assert_allclose(RK45.B, tb["sol"][:6], rtol=1e-15, atol=1e-16)
    assert tb["sol"][6] == 0


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_midpoint_weights_equal_scipy_dense_output_at_one_half(which):
    """y(t0 + h/2) = y0 + h * sum_i c_mid[i] k_i.  scipy: y(t0 + theta h) = y0 + h * theta * sum_i k_i * sum_j P[i,j] theta^j."""
    tb = _tableaus()[which]
    theta = 0.5
    w = np.array([theta * sum(RK45.P[i, j] * theta ** j for j in range(4)) for i in range(7)])
    np.testing.assert_allclose(tb["mid"], w, rtol=1e-12, atol=1e-15)


def _order_sums(b, A, c):
    """Elementary weights of the 17 rooted trees through order 5 (Butcher), paired with 1 / gamma(tree)."""
    b, A, c = np.asarray(b, float), np.asarray(A, float), np.asarray(c, float)
    Ac, Ac2, Ac3 = A @ c, A @ c ** 2, A @ c ** 3
    AAc = A @ Ac
    return [
        (b.sum(), 1.0),
        (b @ c, 1 / 2),
        (b @ c ** 2, 1 / 3), (b @ Ac, 1 / 6),
        (b @ c ** 3, 1 / 4), (b @ (c * Ac), 1 / 8), (b @ Ac2, 1 / 12), (b @ AAc, 1 / 24),
        (b @ c ** 4, 1 / 5), (b @ (c ** 2 * Ac), 1 / 10), (b @ (c * Ac2), 1 / 15), (b @ (c * AAc), 1 / 30), (b @ Ac ** 2, 1 / 20),
        (b @ Ac3, 1 / 20), (b @ (A @ (c * Ac)), 1 / 40), (b @ (A @ Ac2), 1 / 60), (b @ (A @ AAc), 1 / 120),
    ]


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_runge_kutta_order_conditions(which):
    tb = _tableaus()[which]
    A = np.zeros((7, 7))
    for i, row in enumerate(tb["b"]):
        A[i + 1, : len(row)] = row
    c = np.array([0.0] + list(tb["a"]))
    np.testing.assert_allclose(A.sum(1), c, rtol=0, atol=1e-15)  # row-sum condition
    sol = np.array(tb["sol"])
    for got, want in _order_sums(sol, A, c):  # 5th-order solution: all 17 conditions
        assert abs(got - want) < 1e-14
    emb = sol - np.array(tb["err"])          # the embedded solution behind torchdiffeq's error estimate
    cond = _order_sums(emb, A, c)
    for got, want in cond[:8]:               # ... is 4th order
        assert abs(got - want) < 1e-14
    assert max(abs(g - w) for g, w in cond[8:]) > 1e-4   # ... and NOT 5th order, else err would vanish to O(h^6)
    assert abs(sum(tb["err"])) < 1e-16
    # and it is not scipy's embedded pair (same tableau, different 4th-order weights): documented difference
    assert abs(tb["err"][0] - (-RK45.E[0])) > 1e-4


# a stiff-ish nonlinear field on R^6 (decaying, rotating, with a cubic term); time-dependent
_K = np.array([[-8.0, 6.0, 0, 0, 0, 0], [-6.0, -8.0, 0, 0, 0, 0], [0, 0, -1.5, 2.0, 0, 0], [0, 0, -2.0, -1.5, 0, 0], [0, 0, 0, 0, -0.3, 0.7],
               [0.4, 0, 0, 0, -0.7, -0.3]])


def _f_np(t, y):
    return _K @ y - 0.8 * y ** 3 + np.array([np.sin(5 * t), 0, np.cos(3 * t), 0, 1.0, t])


def _f_torch(t, y):
    K = torch.from_numpy(_K)
    tt = t.to(torch.float64)
    forcing = torch.stack([torch.sin(5 * tt), torch.zeros_like(tt), torch.cos(3 * tt), torch.zeros_like(tt), torch.ones_like(tt), tt])
    return K @ y - 0.8 * y ** 3 + forcing


Y0 = np.array([1.0, -0.5, 0.8, 0.3, -1.2, 0.4])


@pytest.mark.parametrize("h", [0.15, 0.11, 0.02, 1e-3])
def test_single_step_and_dense_output_equal_scipy(h):
    """Forced identical step from the same state: y1, f1 (FSAL) and the dense interpolant at theta in (0, 1)."""
    t0 = 0.25
    f0 = _f_np(t0, Y0)
    K = np.empty((7, 6))
    y1_s, f1_s = rk_step(_f_np, t0, Y0, f0, h, RK45.A, RK45.B, RK45.C, K)
    thetas = np.array([0.1, 0.37, 0.5, 0.9, 1.0])
    dense_s = np.stack([Y0 + h * th * (K.T @ np.array([sum(RK45.P[i, j] * th ** j for j in range(4)) for i in range(7)])) for th in thetas])

    y0 = torch.from_numpy(Y0)
    t0t, ht = torch.tensor(t0, dtype=torch.float64), torch.tensor(h, dtype=torch.float64)
    # oracle
    y1_o, f1_o, err_o, k_o = ode_ref._rk_step(_f_torch, y0, torch.from_numpy(f0), t0t, ht, t0t + ht)
    np.testing.assert_allclose(y1_o.numpy(), y1_s, rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(f1_o.numpy(), f1_s, rtol=1e-12, atol=1e-13)
    coef = ode_ref._interp_fit(y0, y1_o, k_o, ht)
    dense_o = np.stack([ode_ref._interp_eval(coef, t0t, t0t + ht, t0t + th * ht).numpy() for th in thetas])
    np.testing.assert_allclose(dense_o, dense_s, rtol=1e-11, atol=1e-12)
    # product: same step forced through Dopri5._step, dense output through advance()
    s = prod.Dopri5(_f_torch, y0, t0t, rtol=1e9, atol=1e9)  # huge tolerance: the forced step is accepted whatever its error
    s.dt = ht
    s._step()
    assert s.accepted == 1 and float(s.t1) == pytest.approx(t0 + h, abs=1e-15)
    np.testing.assert_allclose(s.y0.numpy(), y1_s, rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(s.f0.numpy(), f1_s, rtol=1e-12, atol=1e-13)
    dense_p = np.stack([s.advance(t0t + th * ht).numpy() for th in thetas])
    np.testing.assert_allclose(dense_p, dense_s, rtol=1e-11, atol=1e-12)
    # the error estimate scales as h^5 relative to a half step (4th-order embedded pair): checked at the small steps only
    if h <= 0.02:
        _, _, err_half, _ = ode_ref._rk_step(_f_torch, y0, torch.from_numpy(f0), t0t, ht / 2, t0t + ht / 2)
        ratio = float(err_o.norm() / err_half.norm())
        assert 20 < ratio < 45  # 2^5 = 32


@pytest.mark.parametrize("impl", ["oracle", "product"])
@pytest.mark.parametrize("tol", [1e-5, 1e-7])
def test_adaptive_solve_against_scipy_solve_ivp(impl, tol):
    """Same problem, same rtol = atol, RMS norm in both; reference = scipy at 1e-12.  Integrates DOWN from t = 1 to 0 like the sampler
    (the field is negated so that the backward direction is the stable one)."""
    fn = lambda t, y: -_f_torch(t, y)  # noqa: E731
    f_np = lambda t, y: -_f_np(t, y)  # noqa: E731
    stats = {}
    y0 = torch.from_numpy(Y0)
    t = torch.tensor([1.0, 0.0])
    od = ode_ref.odeint if impl == "oracle" else prod.odeint
    got = od(fn, y0, t, method="dopri5", rtol=tol, atol=tol, stats=stats)[-1].numpy()
    tight = solve_ivp(f_np, (1.0, 0.0), Y0, method="DOP853", rtol=1e-12, atol=1e-13).y[:, -1]
    sci = solve_ivp(f_np, (1.0, 0.0), Y0, method="RK45", rtol=tol, atol=tol)
    e_ours, e_sci = np.abs(got - tight).max(), np.abs(sci.y[:, -1] - tight).max()
    assert e_ours < 50 * tol and e_sci < 50 * tol       # both inside the same global-error band
    assert e_ours < 20 * max(e_sci, tol)                # and ours is not an outlier next to scipy's
    assert stats["nfe"] if "nfe" in stats else True
    n_sci = sci.t.size - 1
    assert 0.4 * n_sci <= stats["accepted"] <= 2.5 * n_sci  # same pair, same norm: comparable step counts


def test_fixed_grid_constructor_and_nfe():
    """torchdiffeq's grid: arange(n)*h + t0 with the last point forced."""
    for h, n in ((0.02, 51), (0.01, 101), (0.1, 11), (0.03, 35)):
        g = ode_ref.fixed_grid(torch.tensor([-1.0, -0.0]), h)
        assert g.numel() == n and float(g[-1]) == 0.0 and float(g[0]) == -1.0
        g2 = prod.fixed_grid(torch.tensor([-1.0, -0.0]), h)
        assert torch.equal(g, g2)
    calls = []

    def f(t, y):
        calls.append(float(t))
        return -y

    ode_ref.odeint(f, torch.ones(2), torch.tensor([1.0, 0.0]), method="euler", options={"step_size": 0.03})
    assert len(calls) == 34 and calls[0] == 1.0 and abs(calls[-1] - 0.01) < 1e-6  # short last step


def test_known_answers():
    """y' = -y and the harmonic oscillator against closed forms; dopri5 global error ~ tolerance; order of the fixed-grid schemes."""
    y0 = torch.tensor([1.0, 2.0], dtype=torch.float64)
    for od in (ode_ref.odeint, prod.odeint):
        out = od(lambda t, y: -y, y0, torch.tensor([0.0, 1.0]), method="dopri5", rtol=1e-8, atol=1e-10)[-1]
        np.testing.assert_allclose(out.numpy(), (y0 * np.exp(-1.0)).numpy(), rtol=1e-7)
        rot = lambda t, y: torch.stack([y[1], -y[0]])  # noqa: E731
        out = od(rot, torch.tensor([1.0, 0.0], dtype=torch.float64), torch.tensor([0.0, 2.0]), method="dopri5", rtol=1e-8, atol=1e-10)[-1]
        np.testing.assert_allclose(out.numpy(), [np.cos(2.0), -np.sin(2.0)], atol=1e-6)
        errs = {}
        for method, order in (("euler", 1), ("midpoint", 2), ("rk4", 4)):
            e = []
            for h in (0.1, 0.05):
                o = od(lambda t, y: -y, y0, torch.tensor([0.0, 1.0], dtype=torch.float64), method=method, options={"step_size": h})[-1]
                e.append(float((o - y0 * np.exp(-1.0)).abs().max()))
            errs[method] = np.log2(e[0] / e[1])
            assert abs(errs[method] - order) < 0.25, (method, errs)


# ----------------------------------------------------------------------------- the other adaptive pairs and --perturb (product only)
def test_bosh3_tableau_equals_scipy_rk23_and_orders():
    from scipy.integrate import RK23

    np.testing.assert_allclose(RK23.C[1:], prod._BS_A[:2], rtol=0, atol=1e-16)
    np.testing.assert_allclose(RK23.A[1, :1], prod._BS_B[0], atol=1e-16)
    np.testing.assert_allclose(RK23.A[2, :2], prod._BS_B[1], atol=1e-16)
    np.testing.assert_allclose(RK23.B, prod._BS_B[2], atol=1e-16)
    np.testing.assert_allclose(-RK23.E, prod._BS_E, atol=1e-16)  # scipy tabulates (embedded - solution), torchdiffeq (solution - embedded)
    # order conditions: solution weights 3rd order, embedded (solution - error) weights 2nd order
    A = np.zeros((4, 4))
    for i, row in enumerate(prod._BS_B):
        A[i + 1, : len(row)] = row
    c = np.array([0.0] + list(prod._BS_A))
    sol = np.array(list(prod._BS_B[2]) + [0.0])
    for got, want in _order_sums(sol, A, c)[:4]:
        assert abs(got - want) < 1e-15
    emb = sol - np.array(prod._BS_E)
    for got, want in _order_sums(emb, A, c)[:2]:
        assert abs(got - want) < 1e-15
    # Heun-Euler: solution = trapezoid (2nd order), embedded = Euler (1st order)
    assert prod._AH_SOL == (0.5, 0.5) and tuple(np.array(prod._AH_SOL) - np.array(prod._AH_E)) == (0.0, 1.0)


@pytest.mark.parametrize("method,stages,tol", [("bosh3", 3, 1e-6), ("adaptive_heun", 1, 1e-5), ("dopri5", 6, 1e-7), ("dopri8", 13, 1e-9)])
def test_adaptive_pairs_solve_and_count(method, stages, tol):
    stats = {}
    y0 = torch.from_numpy(Y0)
    got = prod.odeint(lambda t, y: -_f_torch(t, y), y0, torch.tensor([1.0, 0.0]), method=method, rtol=tol, atol=tol, stats=stats)[-1].numpy()
    tight = solve_ivp(lambda t, y: -_f_np(t, y), (1.0, 0.0), Y0, method="DOP853", rtol=1e-12, atol=1e-13).y[:, -1]
    assert np.abs(got - tight).max() < 300 * tol
    assert stats["nfe"] == 2 + stages * stats["steps"] and stats["accepted"] <= stats["steps"]
    if method == "bosh3":  # same pair, same norm as scipy's RK23: comparable step counts
        n_sci = solve_ivp(lambda t, y: -_f_np(t, y), (1.0, 0.0), Y0, method="RK23", rtol=tol, atol=tol).t.size - 1
        assert 0.4 * n_sci <= stats["accepted"] <= 2.5 * n_sci
    with pytest.raises(NotImplementedError):
        prod.odeint(lambda t, y: -y, y0, torch.tensor([1.0, 0.0]), method="tsit5")


# ----------------------------------------------------------------------------- dopri8: the 13-stage Prince-Dormand 8(7) pair
@functools.lru_cache(None)
def _rooted_trees(n):
    """All rooted trees with n vertices as canonical nested tuples (1, 1, 2, 4, 9, 20, 48, 115 of them for n = 1..8)."""
    if n == 1:
        return [()]

    def parts(total, biggest):
        if total == 0:
            yield []
            return
        for p in range(min(total, biggest), 0, -1):
            for rest in parts(total - p, p):
                yield [p] + rest

    out = set()
    for part in parts(n - 1, n - 1):
        for combo in itertools.product(*[_rooted_trees(p) for p in part]):
            out.add(tuple(sorted(combo)))
    return sorted(out)


def _tree_order(t):
    return 1 + sum(_tree_order(c) for c in t)


def _tree_gamma(t):
    g = _tree_order(t)
    for c in t:
        g *= _tree_gamma(c)
    return g


def _tree_phi(t, A):
    v = np.ones(A.shape[0])
    for c in t:
        v = v * (A @ _tree_phi(c, A))
    return v


def _worst(b, A, orders, theta=1.0):
    return max(abs(b @ _tree_phi(t, A) - theta ** r / _tree_gamma(t)) for r in orders for t in _rooted_trees(r))


def test_dopri8_tableau_satisfies_every_order_condition():
    """torchdiffeq is not installable, so its dopri8 table cannot be diffed; what CAN be checked is the mathematics: a Runge-Kutta method has
    order p iff b . Phi(tau) = 1 / gamma(tau) for every rooted tree tau with at most p vertices.  The 8th-order weights satisfy all 200
    conditions through order 8 and not those of order 9; the embedded weights all 85 through order 7 and not order 8 (so err = O(h^8)); a single
    mistyped digit in any of the ~100 rationals breaks these at the 1e-9 level.  The midpoint weights (ours) satisfy the 17 conditions through
    order 5 at theta = 1/2 on the tableau extended by the first-same-as-last stage."""
    assert [len(_rooted_trees(n)) for n in range(1, 9)] == [1, 1, 2, 4, 9, 20, 48, 115]
    n = 13
    A = np.zeros((n, n))
    for i, row in enumerate(prod._D8_B[:-1]):
        A[i + 1, : len(row)] = row
    c = np.array([0.0] + list(prod._D8_A[:-1]))
    np.testing.assert_allclose(A.sum(1), c, rtol=0, atol=4e-15)
    b8, b7 = np.array(prod._D8_SOL), np.array(prod._D8_EMB)
    assert prod._D8_B[-1] == prod._D8_SOL and prod._D8_A[-1] == 1.0
    assert _worst(b8, A, range(1, 9)) < 4e-15
    assert _worst(b7, A, range(1, 8)) < 4e-15
    assert _worst(b8, A, [9]) > 1e-6 and _worst(b7, A, [8]) > 1e-5
    np.testing.assert_allclose(np.array(prod._D8_E), np.append(b8 - b7, 0.0), rtol=0, atol=0)
    Ae = np.zeros((14, 14))
    Ae[:13, :13] = A
    Ae[13, :13] = b8
    assert _worst(np.array(prod._D8_MID), Ae, range(1, 6), theta=0.5) < 1e-14


def test_dopri8_local_error_is_ninth_order():
    """One forced step of size h and of size h / 2 on the nonlinear field: the local error of an 8th-order method falls by 2^9."""
    y0 = torch.from_numpy(Y0)
    errs = []
    for h in (0.06, 0.03):
        s = prod.Dopri8(_f_torch, y0, torch.tensor(0.0, dtype=torch.float64), 1e-3, 1e-3)
        s.dt = torch.tensor(h, dtype=torch.float64)
        s._step()
        assert s.accepted == 1
        tight = solve_ivp(_f_np, (0.0, h), Y0, method="DOP853", rtol=1e-13, atol=1e-14).y[:, -1]
        errs.append(np.abs(s.y0.numpy() - tight).max())
    assert 8.0 < np.log2(errs[0] / errs[1]) < 10.0, errs


def test_perturb_samples_inside_the_step():
    """options['perturb']: the start-of-step evaluation is one ulp inside the step.  A field that is discontinuous exactly at the grid
    points tells the two modes apart; on a smooth field they agree to rounding."""
    seen = []

    def f(t, y):
        seen.append(float(t))
        return -y

    y0 = torch.ones(2)
    a = prod.odeint(f, y0, torch.tensor([0.0, 1.0]), method="euler", options={"step_size": 0.25})[-1]
    plain = list(seen)
    seen.clear()
    b = prod.odeint(f, y0, torch.tensor([0.0, 1.0]), method="euler", options={"step_size": 0.25, "perturb": True})[-1]
    assert plain == [0.0, 0.25, 0.5, 0.75] and all(p > q for p, q in zip(seen, plain)) and max(p - q for p, q in zip(seen, plain)) < 1e-6
    torch.testing.assert_close(a, b)
    seen.clear()
    prod.odeint(f, y0, torch.tensor([0.0, 1.0]), method="rk4", options={"step_size": 0.5, "perturb": True})
    assert seen[0] > 0.0 and seen[3] < 0.5 and seen[3] > 0.4999  # k1 just after t0, k4 just before t1


def test_fixed_grid_rk4_equals_the_transformers_port_of_torchdiffeq():
    """``transformers`` ships a port of torchdiffeq's fixed-grid solver for one of its models (``RungeKutta4ODESolver`` in
    ``models/qwen2_5_omni``: the 3/8-rule step with its 1/3, 2/3 nodes, the integrate loop over a time grid, linear interpolation of the
    requested times) -- written by other people from the package our restatement could not be diffed against.  Same grid, same field: the
    oracle and the product reproduce it to rounding, including an output time that falls between two grid points and a decreasing time axis."""
    qo = pytest.importorskip("transformers.models.qwen2_5_omni.modeling_qwen2_5_omni")
    y0 = torch.from_numpy(Y0)
    h = 0.0625
    t = torch.tensor([0.0, 0.33, 0.8, 1.0], dtype=torch.float64)
    grid = prod.fixed_grid(t, h)
    theirs = qo.RungeKutta4ODESolver(_f_torch, y0).integrate(grid)
    i = int(0.33 // h)
    at_033 = theirs[i] + (0.33 - grid[i]) / (grid[i + 1] - grid[i]) * (theirs[i + 1] - theirs[i])
    for impl in (prod, ode_ref):
        ours = impl.odeint(_f_torch, y0, t, method="rk4", options={"step_size": h})
        torch.testing.assert_close(ours[-1], theirs[-1], rtol=0, atol=1e-14)
        torch.testing.assert_close(ours[1], at_033, rtol=0, atol=1e-14)
        # the sampler's direction, t: 1 -> 0 (torchdiffeq integrates the negated field on the negated time axis)
        back = impl.odeint(lambda tt, y: -_f_torch(tt, y), y0, torch.tensor([1.0, 0.0], dtype=torch.float64), method="rk4", options={"step_size": h})
        fwd = qo.RungeKutta4ODESolver(lambda tt, y: _f_torch(-tt, y), y0).integrate(prod.fixed_grid(torch.tensor([-1.0, 0.0], dtype=torch.float64), h))
        torch.testing.assert_close(back[-1], fwd[-1], rtol=0, atol=1e-13)

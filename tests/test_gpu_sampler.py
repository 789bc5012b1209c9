"""GPU parity of the solver loop: fused / graph-captured fixed-grid sampling vs the oracle solver driving the oracle DiT,
and vs the product's own generic (unfused) loop.  Tolerance on final latents: rel-L2 <= 1e-3 (SURVEY.md §8c rule 4)."""
from argparse import Namespace

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dit_ref, ode_ref  # checker only


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def _mk(name, dev, **kw):
    from lfm_amd.models import DiT_models

    cfg = dit_ref.DiTCfg.named(name, **kw)
    sd = dit_ref.make_dit_state(cfg, seed=2)
    m = DiT_models[name](img_resolution=32, in_channels=4, **kw)
    m.load_state_dict(sd, strict=True)
    return cfg, sd, m.to(dev).eval()


def test_config1_ditb2_10step_euler_vs_oracle():
    """BASELINE config 1: DiT-B/2, 4x32x32 latents, batch 4, 10-step Euler, no VAE."""
    from lfm_amd.test_flow_latent import sample_from_model

    dev = torch.device("cuda:0")
    cfg, sd, m = _mk("DiT-B/2", dev, num_classes=1, label_dropout=0.0)
    x0 = torch.randn(4, 4, 32, 32, generator=torch.Generator().manual_seed(42))
    ref = ode_ref.odeint(lambda t, x: dit_ref.dit_forward(sd, cfg, t, x), x0, torch.tensor([1.0, 0.0]), method="euler",
                         options={"step_size": 0.1})
    args = Namespace(method="euler", step_size=0.1, perturb=False, compute_nfe=False, cfg_scale=1.0, atol=1e-5, rtol=1e-5)
    got = sample_from_model(m, x0.to(dev), {}, args)
    assert got.shape == (2, 4, 4, 32, 32)
    assert torch.equal(got[0].cpu(), x0)
    assert rel_l2(got[-1], ref[-1]) < 1e-3
    args.fused = False  # generic loop over the same HIP model: same arithmetic, eager launches
    got2 = sample_from_model(m, x0.to(dev), {}, args)
    assert rel_l2(got2[-1], got[-1]) < 1e-5
    args.compute_nfe = True
    _, nfe = sample_from_model(m, x0.to(dev), {}, args)
    assert int(nfe) == 10


def test_cfg_heun_fused_vs_generic_and_oracle():
    """Class-conditional DiT with classifier-free guidance on the Karras grid (config 4 shape, small batch)."""
    from lfm_amd.sampler.karras_sample import karras_sample

    dev = torch.device("cuda:0")
    cfg, sd, m = _mk("DiT-S/2", dev, num_classes=10, label_dropout=0.1)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 4, 32, 32, generator=g)
    x = torch.cat([x, x], 0)
    y = torch.cat([torch.randint(0, 10, (3,), generator=g), torch.full((3,), 10)])
    kw = dict(y=y.to(dev), cfg_scale=1.5)
    common = dict(steps=9, device=dev, clip_denoised=False, sigma_min=1e-5, sigma_max=1.0, s_tmin=0.0, s_tmax=1.0, s_churn=0.0)
    for sampler in ("euler", "heun"):
        fused = karras_sample(m, x.to(dev), model_kwargs=kw, sampler=sampler, fused=True, **common)
        plain = karras_sample(m, x.to(dev), model_kwargs=kw, sampler=sampler, fused=False, **common)
        assert rel_l2(fused, plain) < 1e-5, sampler

        class Oracle:
            def forward_with_cfg(self, t, xx, y=None, cfg_scale=1.0):
                return dit_ref.dit_forward_with_cfg(sd, cfg, t, xx, y, cfg_scale)

        common_cpu = dict(common, device="cpu")
        ref = karras_sample(Oracle(), x, model_kwargs=dict(y=y, cfg_scale=1.5), sampler=sampler, **common_cpu)
        assert rel_l2(fused, ref) < 1e-3, sampler
        assert torch.equal(fused[:3], fused[3:])  # both halves carry the guided trajectory (DiT.py:287)


def test_dopri5_on_hip_model_vs_oracle():
    from lfm_amd.solvers import odeint

    dev = torch.device("cuda:0")
    cfg, sd, m = _mk("DiT-S/2", dev, num_classes=1, label_dropout=0.0)
    x0 = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(3))
    t = torch.tensor([1.0, 0.0])
    sa, sb = {}, {}
    got = odeint(lambda tt, xx: m(tt, xx), x0.to(dev), t.to(dev), method="dopri5", rtol=1e-3, atol=1e-3, stats=sa)
    ref = ode_ref.odeint(lambda tt, xx: dit_ref.dit_forward(sd, cfg, tt, xx), x0, t, method="dopri5", rtol=1e-3, atol=1e-3, stats=sb)
    assert rel_l2(got[-1], ref[-1]) < 2e-3
    assert abs(sa["steps"] - sb["steps"]) <= 1


def test_captured_graph_follows_weight_reload():
    """A captured solver graph holds raw device pointers; reloading weights re-packs them, so the graph must be re-captured."""
    from lfm_amd.solvers import GraphedFixedGrid, torchdiffeq_euler_grid

    dev = torch.device("cuda:0")
    cfg, sd1, m = _mk("DiT-S/2", dev, num_classes=1, label_dropout=0.0)
    sd2 = dit_ref.make_dit_state(cfg, seed=77)
    x0 = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(9)).to(dev)
    ts, dts = torchdiffeq_euler_grid(0.5)
    s = GraphedFixedGrid(m, 2)
    s.set_grid(ts, dts)
    a = s.run(x0).clone()
    m.load_state_dict(sd2, strict=True)
    b = s.run(x0).clone()  # same solver object, new weights
    ref = ode_ref.odeint(lambda t, x: dit_ref.dit_forward(sd2, cfg, t, x), x0.cpu(), torch.tensor([1.0, 0.0]), method="euler",
                         options={"step_size": 0.5})[-1]
    assert rel_l2(b, ref) < 1e-3
    assert rel_l2(a, ref) > 1e-2  # and it really was a different model before


def test_fused_sampler_survives_repeated_calls_and_grid_changes():
    """The captured graph reads the time grid from persistent device buffers: repeated calls, other allocations in between and a
    different grid must all give the eager answer."""
    from lfm_amd.test_flow_latent import sample_from_model

    dev = torch.device("cuda:0")
    cfg, sd, m = _mk("DiT-S/2", dev, num_classes=1, label_dropout=0.0)
    x0 = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(4)).to(dev)
    for h in (0.25, 0.25, 0.125, 0.25):
        junk = [torch.randn(1 << 16, device=dev) for _ in range(8)]  # churn the allocator between calls
        args = Namespace(method="euler", step_size=h, perturb=False, compute_nfe=False, cfg_scale=1.0, atol=1e-5, rtol=1e-5)
        fused = sample_from_model(m, x0, {}, args)[-1]
        args.fused = False
        eager = sample_from_model(m, x0, {}, args)[-1]
        assert rel_l2(fused, eager) < 1e-5, h
        del junk


@pytest.mark.parametrize("name,batch", [("DiT-S/2", 3), ("DiT-L/2", 48)])
def test_cond_table_solve_is_bit_identical(name, batch):
    """Per-grid conditioning tables (lfm_dit_cond_table_build: c, the adaLN modulation and the folded-LayerNorm u / v rows computed once per grid time for the
    unconditional models) against the per-evaluation conditioning: same launches, so the Euler and the Heun solves must agree bit for bit -- for a model
    on the separate-LayerNorm path (small batch) and for DiT-L/2 at a batch that takes the folded path (u / v rows from the table).  Then a grid change and a
    weight reload: the table must follow both."""
    from lfm_amd.solvers import GraphedFixedGrid, torchdiffeq_euler_grid

    dev = torch.device("cuda:0")
    cfg, sd, m = _mk(name, dev, num_classes=1, label_dropout=0.0)
    x0 = torch.randn(batch, 4, 32, 32, generator=torch.Generator().manual_seed(5)).to(dev)
    with_t, without = GraphedFixedGrid(m, batch), GraphedFixedGrid(m, batch)
    assert with_t.use_cond_table
    without.use_cond_table = False
    for h in (0.25, 0.2):
        ts, dts = torchdiffeq_euler_grid(h)
        for s in (with_t, without):
            s.set_grid(ts, dts)
        assert with_t.cond_buf is not None and without.cond_buf is None
        for heun_limit in (0, dts.numel()):
            a = with_t.run(x0, heun_limit=heun_limit).clone()
            b = without.run(x0, heun_limit=heun_limit).clone()
            assert torch.equal(a, b), (name, h, heun_limit)
    if name == "DiT-S/2":
        m.load_state_dict(dit_ref.make_dit_state(cfg, seed=31), strict=True)
        ts, dts = torchdiffeq_euler_grid(0.25)
        for s in (with_t, without):
            s.set_grid(ts, dts)
        assert torch.equal(with_t.run(x0).clone(), without.run(x0).clone())


@pytest.mark.parametrize("kind", ["transposed_view", "expanded", "float64"])
def test_dopri5_device_path_accepts_generic_stage_tensors(kind):
    """odeint() is a torchdiffeq-style API: the derivative may come back as a non-contiguous view, a broadcast (expanded) tensor or another dtype.  The device
    fast path hands raw pointers to lfm_lincomb / lfm_rk_error_norm, so such results must be coerced (or the eager path taken) -- never read as dense fp32."""
    from lfm_amd.solvers import odeint

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(12)
    A = (torch.randn(16, 16, generator=g) * 0.4)
    y0 = torch.randn(8, 16, 16, generator=g)
    c = torch.randn(1, 1, 16, generator=g)

    def field(t, y, A_, c_):
        if kind == "transposed_view":
            return (-(y.transpose(-1, -2).contiguous() @ A_)).transpose(-1, -2)  # same shape, permuted strides
        if kind == "expanded":
            return (c_ * (1.0 + t)).expand(y.shape) - 0.0 * y[..., :1]  # broadcast view (stride 0) of the right shape
        return (-(y.double() @ A_.double()) * (1.0 + t.double()))  # fp64 result for an fp32 state

    t = torch.tensor([0.0, 1.0])
    ref = ode_ref.odeint(lambda tt, yy: field(tt, yy, A, c).to(yy.dtype) if kind == "float64" else field(tt, yy, A, c), y0, t, method="dopri5",
                         rtol=1e-6, atol=1e-6)
    Ad, cd = A.to(dev), c.to(dev)
    got = odeint(lambda tt, yy: field(tt, yy, Ad, cd), y0.to(dev), t.to(dev), method="dopri5", rtol=1e-6, atol=1e-6)
    assert rel_l2(got[-1], ref[-1]) < 1e-5, kind


def test_dopri5_device_path_rejects_steps_like_the_oracle():
    """The adaptive controller's REJECTION branch on the device fast path (one lfm_rk_error_norm read-back per attempt, step shrunk, stage derivatives
    recomputed from the unchanged state): a field whose stiffness jumps by 50x at t = 0.5 forces rejected attempts; attempts, accepted steps and the
    end point must follow the CPU oracle (the DiT configurations' seeded random fields are smooth: 15 / 15 accepted)."""
    from lfm_amd.solvers import odeint

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21)
    y0 = torch.randn(4, 8, 16, generator=g)

    def field(t, y):
        return -y * (1.0 + 50.0 * torch.sigmoid(200.0 * (t - 0.5))) + torch.cos(3.0 * t)

    t = torch.tensor([0.0, 1.0])
    sa, sb = {}, {}
    ref = ode_ref.odeint(field, y0, t, method="dopri5", rtol=1e-5, atol=1e-6, stats=sb)
    got = odeint(field, y0.to(dev), t.to(dev), method="dopri5", rtol=1e-5, atol=1e-6, stats=sa)
    assert sb["steps"] > sb["accepted"], "the oracle did not reject anything: the test field is too smooth"
    assert sa["steps"] > sa["accepted"]
    assert abs(sa["steps"] - sb["steps"]) <= 2 and abs(sa["accepted"] - sb["accepted"]) <= 2, (sa, sb)
    assert rel_l2(got[-1], ref[-1]) < 1e-4

"""Two HIP streams in flight on one GPU (the default of bench.py since round 5, and what the multi-rank driver does by design: the RCCL all-gather of a
batch runs on a side stream under the next batch's solve, lfm_amd/test_flow_latent_ddp.py::GatherPipeline).  A sampling job's batches are independent
(/root/reference/test_flow_latent_ddp.py:128-146), so co-scheduling is legal -- PROVIDED every evaluation stays bit-identical to its solo result.

Round 4 found that it did not (~4 % of co-scheduled DiT-L/2 evaluations on the folded-LayerNorm path differed by 1-2 fp16 ulp in the fc1 activation);
round 5 root-caused it (profiles/r05_cosched_root_cause.txt: v_pk_fma_f32 with op_sel reads an operand as 0.0 in lanes 48-63 when a foreign wave shares
the SIMD; csrc/common.h fma_v).  This file is the regression guard on hardware: >= 200 co-scheduled evaluations, 0 may differ.  The static guard (no
op_sel'd packed-fp32 instruction in the library) is tests/test_host_logic.py::test_no_packed_fp32_op_sel.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dit_ref, ode_ref, vae_ref  # checkers only


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def dit_l2(dev):
    from lfm_amd.models import DiT_models

    kw = dict(num_classes=1, label_dropout=0.0)
    cfg = dit_ref.DiTCfg.named("DiT-L/2", **kw)
    sd = dit_ref.make_dit_state(cfg, seed=0)
    m = DiT_models["DiT-L/2"](img_resolution=32, in_channels=4, **kw)
    m.load_state_dict(sd, strict=True)
    return cfg, sd, m.to(dev).eval()


def test_cosched_evaluations_are_bit_identical_to_solo(dev, dit_l2):
    """DiT-L/2, batch 64, folded path (the bench's evaluation): 240 evaluations that share the GPU with (a) a twin's evaluations on a second stream -- two
    batches in flight, bench.py's default -- and (b) additionally a side stream pumping device-to-device and device-to-host copies, standing in for the
    all-gather + host copy of the previous batch.  Every output and the whole workspace of every compared evaluation must equal the solo run's."""
    from lfm_amd.solvers import concurrency_twin

    _, _, m = dit_l2
    twin = concurrency_twin(m)
    x = torch.randn(64, 4, 32, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    x2 = torch.randn(64, 4, 32, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    t = torch.tensor(0.5, device=dev)
    ref = m(t, x).clone()
    assert torch.equal(m(t, x), ref), "solo evaluations must repeat bit for bit"
    ws_ref = m._ws[1].clone()
    twin(t, x2)
    torch.cuda.synchronize()
    sa, sb, sc = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    blob = torch.empty(48 << 20, dtype=torch.uint8, device=dev)  # ~ the u8 image block of four ranks' batches
    blob2 = torch.empty_like(blob)
    host = torch.empty(12 << 20, dtype=torch.uint8).pin_memory()
    differ, compared = 0, 0
    for rep in range(80):
        cur = torch.cuda.current_stream(dev)
        for s in (sa, sb, sc):
            s.wait_stream(cur)
        with torch.cuda.stream(sb):
            for _ in range(4):
                twin(t, x2)
        if rep >= 40:
            with torch.cuda.stream(sc):
                for _ in range(6):
                    blob2.copy_(blob, non_blocking=True)
                    host.copy_(blob2[: host.numel()], non_blocking=True)
        outs = []
        with torch.cuda.stream(sa):
            for _ in range(3):
                outs.append(m(t, x).clone())
        torch.cuda.synchronize()
        for o in outs:
            compared += 1
            differ += int(not torch.equal(o, ref))
        differ += int(not torch.equal(m._ws[1], ws_ref))  # the last evaluation's intermediate buffers too
    assert compared >= 200
    assert differ == 0, f"{differ} of {compared} co-scheduled evaluations differ from the solo result"


def test_headline_config_end_to_end_vs_oracle_subsampled(dev, dit_l2):
    """BASELINE config 2 end to end, as bench.py times it: DiT-L/2, 64 latents, 50 Euler steps on the torchdiffeq grid (captured graph, fused update,
    per-grid conditioning table), VAE decode, uint8 -- through TWO lanes in flight.  The oracle re-integrates images 0, 31 and 63 only (the unconditional
    model treats the images of a batch independently: reference run_sampling, test_flow_latent.py:161-194): latents <= 1e-3 rel-L2, uint8 images within
    one step on >= 99.9 % of the pixels.  Both lanes must agree bit for bit."""
    from lfm_amd.autoencoder import AutoencoderKL, images_to_uint8
    from lfm_amd.solvers import GraphedFixedGrid, concurrency_twin, torchdiffeq_euler_grid

    cfg, sd, m = dit_l2
    vsd = vae_ref.make_vae_state(seed=0)
    vae = AutoencoderKL()
    vae.load_state_dict(vsd)
    vae = vae.to(dev)
    x0 = torch.randn(64, 4, 32, 32, generator=torch.Generator().manual_seed(42))
    ts, dts = torchdiffeq_euler_grid(0.02)
    assert dts.numel() == 50
    lanes = []
    for mod, va in ((m, vae), (concurrency_twin(m), concurrency_twin(vae))):
        sv = GraphedFixedGrid(mod, 64)
        sv.set_grid(ts, dts)
        lanes.append((sv, va, torch.cuda.Stream(dev)))
    res = []
    for sv, va, st in lanes:
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            lat = sv.run(x0.to(dev, non_blocking=True)).clone()
            res.append((lat, images_to_uint8(va.decode(lat / 0.18215).sample)))
    torch.cuda.synchronize()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), "two lanes in flight: same latents, same images, bit for bit"
    lat, u8 = res[0]
    pick = [0, 31, 63]
    ref_lat = ode_ref.odeint(lambda tt, xx: dit_ref.dit_forward(sd, cfg, tt, xx), x0[pick], torch.tensor([1.0, 0.0]), method="euler",
                             options={"step_size": 0.02})[-1]
    e = float((lat[pick].cpu().double() - ref_lat.double()).norm() / ref_lat.double().norm())
    assert e < 1e-3, e
    ref_img = vae_ref.vae_decode(vsd, ref_lat / 0.18215)
    ref_u8 = ((ref_img + 1) / 2).clamp(0, 1).mul(255).to(torch.uint8).permute(0, 2, 3, 1)  # test_flow_latent_ddp.py:131-135 (truncation)
    d = (u8[pick].cpu().int() - ref_u8.int()).abs()
    assert float((d <= 1).float().mean()) >= 0.999, float((d <= 1).float().mean())
    assert int(d.max()) <= 3

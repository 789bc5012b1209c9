"""GPU parity: HIP VAE decoder (C ABI) vs the fp32 oracle restatement, identical seeded weights + latents.
Tolerance: rel-L2 <= 5e-3 on the decoded image (30 fp16 layers with GroupNorm re-normalisation)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vae_ref  # checker only


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("N,R,chunk", [(2, 8, 16), (3, 16, 2), (1, 32, 16)])
def test_vae_decode_matches_oracle(N, R, chunk):
    from lfm_amd.autoencoder import AutoencoderKL

    dev = torch.device("cuda:0")
    sd = vae_ref.make_vae_state(seed=3)
    vae = AutoencoderKL(decode_chunk=chunk)
    vae.load_state_dict(sd, strict=True)
    vae = vae.to(dev)
    z = torch.randn(N, 4, R, R, generator=torch.Generator().manual_seed(N + R)) * 1.5
    ref = vae_ref.vae_decode(sd, z)
    got = vae.decode(z.to(dev)).sample
    assert got.shape == (N, 3, 8 * R, 8 * R)
    assert float(ref.abs().mean()) > 1e-2
    assert rel_l2(got, ref) < 5e-3
    assert torch.equal(got, vae.decode(z.to(dev)).sample)  # two-stage GroupNorm statistics: bit-for-bit repeatable


def test_images_to_uint8():
    from lfm_amd.autoencoder import images_to_uint8

    x = torch.randn(2, 3, 16, 24, generator=torch.Generator().manual_seed(0)) * 1.2
    ref = (torch.clamp((x + 1) / 2, 0, 1) * 255).permute(0, 2, 3, 1).to(torch.uint8)
    got = images_to_uint8(x.cuda()).cpu()
    assert torch.equal(got, ref)
    # rounding mode = the single-process script's torchvision.utils.save_image conversion, bit-exact against the host restatement
    from lfm_amd.io_formats import to_uint8_rounding, to_uint8_truncating

    assert torch.equal(ref, to_uint8_truncating(x))
    x01 = torch.clamp((x + 1) / 2, 0, 1)
    assert torch.equal(images_to_uint8(x.cuda(), rounding=True).cpu(), to_uint8_rounding(x01))
    edge = torch.tensor([-1.0, 1.0, 2 * 100.5 / 255 - 1, 2 * 254.5 / 255 - 1, 0.0, 5.0, -5.0, 2 * 0.5 / 255 - 1]).reshape(1, 1, 1, 8).expand(1, 3, 1, 8).contiguous()
    assert torch.equal(images_to_uint8(edge.cuda(), rounding=True).cpu(), to_uint8_rounding(torch.clamp((edge + 1) / 2, 0, 1)))


@pytest.mark.parametrize("N,S,chunk", [(2, 64, 16), (3, 128, 2), (1, 256, 16)])
def test_vae_encode_matches_oracle(N, S, chunk):
    """Encoder half (train_flow_latent.py:143, downstream_tasks/test_flow_latent_inpainting.py:146): moments of the latent
    distribution vs the fp32 restatement; quant_conv is folded into conv_out on the product side.  Then the round trip
    decode(mode) runs through both halves on the GPU and is compared with the oracle's."""
    from lfm_amd.autoencoder import AutoencoderKL

    dev = torch.device("cuda:0")
    sd = vae_ref.make_vae_state(seed=3, with_encoder=True)
    vae = AutoencoderKL(decode_chunk=chunk, with_encoder=True)
    vae.load_state_dict(sd, strict=True)
    vae = vae.to(dev)
    x = torch.randn(N, 3, S, S, generator=torch.Generator().manual_seed(N + S)).clamp(-1, 1)
    ref = vae_ref.vae_encode_moments(sd, x)
    dist = vae.encode(x.to(dev)).latent_dist
    assert dist.parameters.shape == (N, 8, S // 8, S // 8)
    assert float(ref.abs().mean()) > 1e-2
    assert rel_l2(dist.parameters, ref) < 5e-3
    assert torch.equal(dist.parameters, vae.encode(x.to(dev)).latent_dist.parameters)  # deterministic
    z = dist.sample(generator=torch.Generator(dev).manual_seed(1))
    assert z.shape == (N, 4, S // 8, S // 8) and bool(torch.isfinite(z).all())
    img = vae.decode(dist.mode()).sample
    ref_img = vae_ref.vae_decode(sd, ref[:, :4])
    assert rel_l2(img, ref_img) < 1e-2


@pytest.mark.parametrize("N,H,W,Cin,Cout,mode", [(2, 48, 48, 128, 128, 0), (2, 32, 64, 64, 128, 0), (1, 64, 32, 256, 256, 0), (3, 16, 16, 192, 128, 0),
                                                  (2, 32, 32, 128, 384, 1), (1, 96, 32, 64, 128, 1)])
def test_halo_conv_matches_torch_and_the_implicit_gemm(N, H, W, Cin, Cout, mode):
    """conv_halo_kernel.h (the VAE decoder's 128-channel 3x3 convolutions) through the C ABI: against the fp32 torch convolution of the same
    fp16-rounded operands, and against the implicit-GEMM kernel on identical inputs (same products, another summation order: within 2 fp16
    ulps of the output range).  Shapes cover border and interior tiles, the XCD tile remap on and off, 2 / 4 / 6 / 8 channel quarters,
    one to three 128-channel output blocks, with and without the fused residual; mode 1 = the nearest-2x upsample fused into the gather
    (H, W are the output size)."""
    import torch.nn.functional as F

    from lfm_amd import hip

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N * 1000 + H + Cin)
    L = hip.lib()
    x = torch.randn(N, Cin, H >> mode, W >> mode, generator=g).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).half()
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(N, Cout, H, W, generator=g).half()
    xin = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(dev)
    bd = b.to(dev)
    rd = res.permute(0, 2, 3, 1).contiguous().to(dev)

    def run(flags, resid):
        out = torch.full((N * H * W, Cout), float("nan"), dtype=torch.float16, device=dev)
        hip.gemm_select(flags << 4)
        try:
            hip.check(L.lfm_conv3x3_f16(hip.ptr(xin), hip.ptr(wp), hip.ptr(bd), hip.ptr(resid), hip.ptr(out), N, H, W, Cin, Cout, mode,
                                        hip.stream_ptr()), "conv")
        finally:
            hip.gemm_select(0)
        return out.reshape(N, H, W, Cout).permute(0, 3, 1, 2).float().cpu()

    xs = F.interpolate(x.float(), scale_factor=2, mode="nearest") if mode else x.float()
    ref = F.conv2d(xs, w.float(), b, padding=1)
    for resid, r in ((None, ref), (rd, ref + res.float())):
        halo, gemm = run(16777216, resid), run(8388608, resid)  # flags: the halo kernel at any size / the implicit GEMM
        assert torch.isfinite(halo).all()
        assert rel_l2(halo, r) < 1e-3
        assert float((halo - gemm).abs().max()) <= 2 * 2.0 ** -10 * float(r.abs().max())

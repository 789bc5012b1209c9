"""Latent-flow image inpainting sampler (drop-in for /root/reference/downstream_tasks/test_flow_latent_inpainting.py).

Per batch (reference :141-160): encode the masked image with the f8 VAE, resize the mask to the latent grid, concatenate both to the
state (4 + 4 + 1 = 9 input channels of the origin-ADM UNet), integrate the flow from noise, decode, and paste the generated pixels into
the hole:  out = fake * mask + (1 - mask) * image  (all in [0, 1]).  Every heavy step -- VAE encode, UNet velocity field, VAE decode --
runs on the HIP path; the dataset / checkpoint plumbing of the reference's ``sample_and_test`` stays host code."""
import argparse

import torch
import torch.nn.functional as F

from . import ADAPTIVE_SOLVER, FIXER_SOLVER, WrapperCondFlow, sample_from_model  # noqa: F401

to_range_0_1 = lambda x: (x + 1.0) / 2.0  # noqa: E731  (reference :95)


@torch.no_grad()
def inpaint_batch(model_cond, first_stage_model, image, mask, masked_image, args, z_0=None, generator=None, sample_posterior=True):
    """One iteration of the reference's loop (:141-160).  image, masked_image: [N,3,S,S] in [-1,1]; mask: [N,1,S,S] in {-1,+1} with +1 = hole
    (dataset convention :44-53).  Returns (composited image in [0,1], generated latent)."""
    dist = first_stage_model.encode(masked_image).latent_dist
    c = (dist.sample(generator=generator) if sample_posterior else dist.mode().clone()).mul_(args.scale_factor)  # reference: .sample()
    cc = F.interpolate(mask, size=c.shape[-2:])  # nearest, as the reference's default
    model_cond.cond = torch.cat((c, cc), dim=1)
    if z_0 is None:
        z_0 = torch.randn(image.size(0), 4, args.image_size // 8, args.image_size // 8, device=image.device, generator=generator)
    fake_sample = sample_from_model(model_cond, z_0, args)[-1]
    fake_image = first_stage_model.decode(fake_sample / args.scale_factor).sample
    fake_image, m, img = to_range_0_1(fake_image), to_range_0_1(mask), to_range_0_1(image)
    return fake_image * m + (1 - m) * img, fake_sample


def build_parser():
    """The reference's argparse surface (:165-226), same names and defaults."""
    p = argparse.ArgumentParser("latent-flow inpainting")
    p.add_argument("--seed", type=int, default=1024)
    p.add_argument("--compute_fid", action="store_true", default=False)
    p.add_argument("--epoch_id", type=int, default=500)
    p.add_argument("--image_size", type=int, default=256)
    p.add_argument("--num_in_channels", type=int, default=9)
    p.add_argument("--num_out_channels", type=int, default=4)
    p.add_argument("--nf", type=int, default=256)
    p.add_argument("--centered", action="store_false", default=True)
    p.add_argument("--resamp_with_conv", type=bool, default=True)
    p.add_argument("--num_res_blocks", type=int, default=2)
    p.add_argument("--num_heads", type=int, default=4)
    p.add_argument("--num_head_upsample", type=int, default=-1)
    p.add_argument("--num_head_channels", type=int, default=-1)
    p.add_argument("--attn_resolutions", nargs="+", type=int, default=(16, 8))
    p.add_argument("--ch_mult", nargs="+", type=int, default=(1, 2, 3, 4))
    p.add_argument("--dropout", type=float, default=0.0)
    p.add_argument("--num_classes", type=int, default=None)
    p.add_argument("--use_scale_shift_norm", type=bool, default=True)
    p.add_argument("--resblock_updown", type=bool, default=False)
    p.add_argument("--use_new_attention_order", type=bool, default=False)
    p.add_argument("--scale_factor", type=float, default=0.18215)
    p.add_argument("--exp", default="latent_kl_exp1")
    p.add_argument("--real_img_dir", default="./pytorch_fid/cifar10_train_stat.npy")
    p.add_argument("--dataset", default="celeba_256")
    p.add_argument("--num_timesteps", type=int, default=200)
    p.add_argument("--batch_size", type=int, default=50)
    p.add_argument("--atol", type=float, default=1e-5)
    p.add_argument("--rtol", type=float, default=1e-5)
    p.add_argument("--method", type=str, default="dopri5", choices=["dopri5", "dopri8", "adaptive_heun", "bosh3", "euler", "midpoint", "rk4"])
    p.add_argument("--step_size", type=float, default=0.01)
    p.add_argument("--perturb", action="store_true", default=False)
    p.add_argument("--pretrained_autoencoder_ckpt", type=str, default="../stabilityai/sd-vae-ft-mse")
    p.add_argument("--random_weights", action="store_true", help="synthetic weights and synthetic masks instead of checkpoints / datasets")
    return p


def build_models(args, device):
    from ..autoencoder import AutoencoderKL
    from ..io_formats import load_state_dict_file
    from ..models import get_flow_model
    from ..test_flow_latent import dezero_

    args.layout = False
    torch.manual_seed(args.seed)
    model = get_flow_model(args)
    if args.random_weights:
        dezero_(model)
        vae = AutoencoderKL.from_random(seed=args.seed, with_encoder=True)
    else:
        path = "./saved_info/latent_flow_inpaint/{}/{}/model_{}.pth".format(args.dataset, args.exp, args.epoch_id)
        model.load_state_dict(load_state_dict_file(path), strict=True)
        vae = AutoencoderKL.from_pretrained(args.pretrained_autoencoder_ckpt, with_encoder=True)
    return model.to(device).eval(), vae.to(device)


def synthetic_batch(n, size, device, seed=0):
    """Stand-in for CustomizedInpaintingEvalDataset (:27-55) when no dataset is on disk: smooth random images and box masks."""
    g = torch.Generator().manual_seed(seed)
    img = F.interpolate(torch.rand(n, 3, size // 16, size // 16, generator=g), size=(size, size), mode="bilinear", align_corners=False) * 2 - 1
    mask01 = torch.zeros(n, 1, size, size)
    for i in range(n):
        y0, x0 = [int(v) for v in torch.randint(size // 8, size // 2, (2,), generator=g)]
        mask01[i, :, y0:y0 + size // 3, x0:x0 + size // 3] = 1.0
    masked = (1 - mask01) * ((img + 1) / 2)  # the dataset masks in [0,1] space, then maps everything to [-1,1]
    return img.to(device), (mask01 * 2 - 1).to(device), (masked * 2 - 1).to(device)


def main(argv=None):
    args = build_parser().parse_args(argv)
    torch.set_grad_enabled(False)
    device = torch.device("cuda:0")
    model, vae = build_models(args, device)
    model_cond = WrapperCondFlow(model, cond=None)
    if not args.random_weights:
        raise SystemExit("the reference's dataset classes (./gt_celeb, masks_val_256_small_eval) are not part of this package: "
                         "call inpaint_batch() from your own loader, or run with --random_weights for a synthetic batch")
    image, mask, masked = synthetic_batch(args.batch_size, args.image_size, device, seed=args.seed)
    out, _ = inpaint_batch(model_cond, vae, image, mask, masked, args)
    print(f"inpainted {out.shape[0]} images at {out.shape[-1]}x{out.shape[-1]}: range [{float(out.min()):.3f}, {float(out.max()):.3f}]")


if __name__ == "__main__":
    main()

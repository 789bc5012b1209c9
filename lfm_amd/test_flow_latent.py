"""Single-process sampling driver (drop-in for /root/reference/test_flow_latent.py).

Keeps the reference's names and flags: ``ADAPTIVE_SOLVER``, ``FIXER_SOLVER``, ``NFECount``, ``sample_from_model``,
``sample_from_model_with_fixed_step_solver`` (which the reference's own CLI mis-spells at :188 -- fixed here), the full
argparse surface (:303-409) and the four modes (--compute_fid / --compute_nfe / --measure_time / default grid).

Differences, all documented in DESIGN.md "Reference quirks": device literals follow the model's device; ``module.`` is
stripped from checkpoint keys only when present; CFG with num_classes in {None, 1} raises instead of silently halving the
batch; FID needs torchvision/Inception weights that are not available offline, so --compute_fid writes the images and
says so.
"""
import argparse
import math
import os
import time

import torch
from torch import nn

from .models import create_network
from .sampler.karras_sample import karras_sample
from .sampler.random_util import get_generator
from .solvers import (ADAPTIVE_SOLVER, FIXER_SOLVER, fused_fixed_grid_available, odeint, sample_torchdiffeq_euler_fused)

__all__ = ["ADAPTIVE_SOLVER", "FIXER_SOLVER", "NFECount", "sample_from_model", "sample_from_model_with_fixed_step_solver",
           "run_sampling", "build_parser", "main"]


class NFECount(nn.Module):
    """Counts velocity-field evaluations (reference test_flow_latent.py:31-39)."""

    def __init__(self, model):
        super().__init__()
        self.model = model
        self.register_buffer("nfe", torch.tensor(0.0))

    def __call__(self, t, x, *args, **kwargs):
        self.nfe += 1.0
        return self.model(t, x, *args, **kwargs)

    def forward_with_cfg(self, t, x, *args, **kwargs):
        self.nfe += 1.0
        return self.model.forward_with_cfg(t, x, *args, **kwargs)


def sample_from_model(model, x_0, model_kwargs, args):
    """Reference test_flow_latent.py:42-76: integrate dx/dt = v(t, x) from t=1 (noise) to t=0 (data).
    Returns the [2, N, C, h, w] trajectory endpoints (callers take [-1]), or (traj, nfe) under --compute_nfe."""
    if args.method in ADAPTIVE_SOLVER:
        options = {"dtype": torch.float64}
    else:
        options = {"step_size": args.step_size, "perturb": args.perturb}
    count = getattr(args, "compute_nfe", False)
    use_cfg = getattr(args, "cfg_scale", 1.0) > 1.0

    fused_ok = (args.method == "euler" and not count and not getattr(args, "perturb", False)
                and fused_fixed_grid_available(model, x_0) and getattr(args, "fused", True))
    if fused_ok:
        kw = dict(model_kwargs)
        if not use_cfg:
            kw.pop("cfg_scale", None)
        x1 = sample_torchdiffeq_euler_fused(model, x_0, args.step_size, kw)
        return torch.stack([x_0, x1], 0)

    if count:
        model = NFECount(model)
        model.nfe = model.nfe.to(x_0.device)  # only the counter moves: .to() on the wrapper would walk the wrapped model
    t = torch.tensor([1.0, 0.0], device=x_0.device)

    def denoiser(t, x):
        if use_cfg:
            return model.forward_with_cfg(t, x, **model_kwargs)
        return model(t, x, **{k: v for k, v in model_kwargs.items() if k != "cfg_scale"})

    traj = odeint(denoiser, x_0, t, method=args.method, atol=args.atol, rtol=args.rtol,
                  options={k: v for k, v in options.items() if k in ("step_size", "perturb")})
    if count:
        return traj, model.nfe
    return traj


def sample_from_model_with_fixed_step_solver(model, x, model_kwargs, generator, args):
    """Reference test_flow_latent.py:79-97 (Karras-style linear grid, sigma in [1e-5, 1])."""
    return karras_sample(model, x, steps=args.num_steps, model_kwargs=model_kwargs, device=x.device, clip_denoised=False,
                         sigma_min=1e-5, sigma_max=1.0, s_tmin=0.0, s_tmax=1.0, s_churn=0.0, sampler=args.method, rho=1.0,
                         ts=range(0, args.num_steps, 15), generator=generator,
                         heun_reference_quirk=getattr(args, "heun_reference_quirk", True))


def make_model_kwargs(args, x, generator, device, cls_index=None):
    """Labels + classifier-free-guidance doubling (reference test_flow_latent.py:163-183)."""
    n = x.shape[0]
    if args.num_classes in [None, 1]:
        if args.cfg_scale > 1.0:
            raise ValueError("--cfg_scale > 1 needs a class-conditional model (num_classes > 1); the reference silently "
                             "discards half the batch in this configuration (test_flow_latent.py:56-57,163-164,190-191)")
        return x, {}
    if cls_index is None:
        y = generator.randint(0, args.num_classes, (n,), device=device).to(device)
    else:
        y = torch.full((n,), int(cls_index), device=device, dtype=torch.long)
    if args.cfg_scale > 1.0:
        x = torch.cat([x, x], 0)
        y_null = torch.full((n,), args.num_classes, device=device, dtype=torch.long) if "DiT" in args.model_type else torch.zeros_like(y)
        return x, dict(y=torch.cat([y, y_null], 0), cfg_scale=args.cfg_scale)
    return x, dict(y=y)


def run_sampling(model, first_stage_model, args, num_samples, generator, device, cls_index=None, x=None):
    """One batch of the hot path: noise -> ODE solve -> (drop null half) -> VAE decode (reference :161-194)."""
    if x is None:
        x = generator.randn(num_samples, 4, args.image_size // 8, args.image_size // 8).to(device)
    x, model_kwargs = make_model_kwargs(args, x, generator, device, cls_index)
    if not args.use_karras_samplers:
        fake_sample = sample_from_model(model, x, model_kwargs, args)[-1]
    else:
        fake_sample = sample_from_model_with_fixed_step_solver(model, x, model_kwargs, generator, args)
    if args.cfg_scale > 1.0:
        fake_sample, _ = fake_sample.chunk(2, dim=0)
    if first_stage_model is None:
        return fake_sample
    return first_stage_model.decode(fake_sample / args.scale_factor).sample


def make_lanes(model, vae, device, n_lanes):
    """[(model, vae, stream)] x n_lanes for batches in flight on one GPU: lane 0 is the model itself, the others are concurrency twins (same packed weights, own
    workspace / solver buffers / captured graphs).  Every lane stream first waits for the stream the weights were loaded and packed on.  Batches of a sampling job
    are independent (reference test_flow_latent_ddp.py:128-146) and a lane's result is bit-identical to the one-lane result (tests/test_gpu_cosched.py)."""
    from .solvers import concurrency_twin

    cur = torch.cuda.current_stream(device)
    lanes = []
    for k in range(max(1, n_lanes)):
        st = torch.cuda.Stream(device)
        st.wait_stream(cur)
        lanes.append((model if k == 0 else concurrency_twin(model), vae if k == 0 or vae is None else concurrency_twin(vae), st))
    return lanes


def load_checkpoint(model, path, device, trust_checkpoint=False):
    """``model_{epoch}.pth`` (flat state_dict of an accelerate/DDP-wrapped model, train_flow_latent.py:211-214: ``module.`` stripped when
    present) or ``content.pth`` (train_flow_latent.py:196-203: the weights sit under ``model_dict``)."""
    from .io_formats import load_state_dict_file

    model.load_state_dict(load_state_dict_file(path, map_location=device, trust_checkpoint=trust_checkpoint), strict=True)


def build_parser():
    p = argparse.ArgumentParser("flow-matching parameters")
    p.add_argument("--generator", type=str, default="determ", choices=["dummy", "determ", "determ-indiv", "device"])
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--in_flight", type=int, default=0, help="(_ddp driver) batches in flight per GPU on alternating HIP streams; 0 = 2 on a GPU (bit-identical "
                   "to 1: the batches of a job are independent), 1 = strictly one after the other")
    p.add_argument("--compute_fid", action="store_true", default=False)
    p.add_argument("--compute_nfe", action="store_true", default=False)
    p.add_argument("--measure_time", action="store_true", default=False)
    p.add_argument("--epoch_id", type=int, default=1000)
    p.add_argument("--n_sample", type=int, default=50000)
    p.add_argument("--model_type", type=str, default="adm")
    p.add_argument("--image_size", type=int, default=32)
    p.add_argument("--f", type=int, default=8)
    p.add_argument("--scale_factor", type=float, default=0.18215)
    p.add_argument("--num_in_channels", type=int, default=3)
    p.add_argument("--num_out_channels", type=int, default=3)
    p.add_argument("--nf", type=int, default=256)
    p.add_argument("--centered", action="store_false", default=True)
    p.add_argument("--resamp_with_conv", type=bool, default=True)
    p.add_argument("--num_res_blocks", type=int, default=2)
    p.add_argument("--num_heads", type=int, default=4)
    p.add_argument("--num_head_upsample", type=int, default=-1)
    p.add_argument("--num_head_channels", type=int, default=-1)
    p.add_argument("--attn_resolutions", nargs="+", type=int, default=(16,))
    p.add_argument("--ch_mult", nargs="+", type=int, default=(1, 2, 2, 2))
    p.add_argument("--label_dim", type=int, default=0)
    p.add_argument("--augment_dim", type=int, default=0)
    p.add_argument("--dropout", type=float, default=0.0)
    p.add_argument("--num_classes", type=int, default=None)
    p.add_argument("--label_dropout", type=float, default=0.0)
    p.add_argument("--cfg_scale", type=float, default=1.0)
    p.add_argument("--layout", action="store_true")
    p.add_argument("--use_origin_adm", action="store_true")
    p.add_argument("--use_scale_shift_norm", type=bool, default=True)
    p.add_argument("--resblock_updown", type=bool, default=False)
    p.add_argument("--use_new_attention_order", type=bool, default=False)
    p.add_argument("--pretrained_autoencoder_ckpt", type=str, default="stabilityai/sd-vae-ft-mse")
    p.add_argument("--output_log", type=str, default="")
    p.add_argument("--exp", default="experiment_cifar_default")
    p.add_argument("--real_img_dir", default="./pytorch_fid/cifar10_train_stat.npy")
    p.add_argument("--dataset", default="cifar10")
    p.add_argument("--num_steps", type=int, default=40)
    p.add_argument("--batch_size", type=int, default=200)
    p.add_argument("--use_karras_samplers", action="store_true", default=False)
    p.add_argument("--atol", type=float, default=1e-5)
    p.add_argument("--rtol", type=float, default=1e-5)
    p.add_argument("--method", type=str, default="dopri5",
                   choices=["dopri5", "dopri8", "adaptive_heun", "bosh3", "euler", "midpoint", "rk4", "heun", "multistep", "stochastic", "dpm"])
    p.add_argument("--step_size", type=float, default=0.01)
    p.add_argument("--perturb", action="store_true", default=False)
    p.add_argument("--num_proc_node", type=int, default=1)
    p.add_argument("--num_process_per_node", type=int, default=1)
    p.add_argument("--node_rank", type=int, default=0)
    p.add_argument("--local_rank", type=int, default=0)
    p.add_argument("--master_address", type=str, default="127.0.0.1")
    p.add_argument("--master_port", type=str, default="6000")
    # additions (not in the reference)
    p.add_argument("--random_weights", action="store_true", help="synthetic weights instead of ./saved_info checkpoints (benchmarking)")
    p.add_argument("--heun_reference_quirk", type=int, default=1, help="1: corrector only on intervals < 39 as the reference does")
    p.add_argument("--save_dir", type=str, default=None)
    p.add_argument("--fid_feature_extractor", type=str, default="", help="pkg.module:factory -- factory(device) returns f(uint8 NHWC batch) -> "
                   "[n, 2048] pool3 features; with it --compute_fid reduces the statistics on the GPUs (lfm_amd/fid.py) instead of writing JPEGs")
    p.add_argument("--trust_checkpoint", action="store_true", help="allow the full (unsafe) unpickler for checkpoints the safe loader rejects")
    return p


def dezero_(model, seed=1234, std=0.02):
    """Random-init models output exactly 0 (adaLN-Zero); re-draw all-zero tensors so synthetic runs do real work."""
    g = torch.Generator().manual_seed(seed)
    for _, p in sorted(model.named_parameters()):
        if p.numel() and not bool(p.any()):
            p.data.copy_(torch.randn(p.shape, generator=g) * std)
    return model


def build_models(args, device):
    from .autoencoder import AutoencoderKL

    torch.manual_seed(args.seed)
    model = create_network(args)
    if args.random_weights:
        dezero_(model)
        vae = AutoencoderKL.from_random(seed=args.seed)
    else:
        load_checkpoint(model, "./saved_info/latent_flow/{}/{}/model_{}.pth".format(args.dataset, args.exp, args.epoch_id), "cpu",
                        trust_checkpoint=getattr(args, "trust_checkpoint", False))
        vae = AutoencoderKL.from_pretrained(args.pretrained_autoencoder_ckpt)
    return model.to(device).eval(), vae.to(device)


def save_images_uint8(img_u8, save_dir, start_index, stride=1, offset=0):
    """One JPEG per image, named by the reference's global index j * world + rank + total (test_flow_latent.py:268, _ddp:138)."""
    from .io_formats import save_indexed_jpegs

    save_indexed_jpegs(img_u8.cpu(), save_dir, start_index, world_size=stride, rank=offset)


def main(argv=None):
    from .autoencoder import images_to_uint8

    args = build_parser().parse_args(argv)
    args.world_size = args.num_proc_node * args.num_process_per_node
    torch.set_grad_enabled(False)
    device = torch.device("cuda:{}".format(args.local_rank))
    torch.cuda.set_device(device)
    model, vae = build_models(args, device)
    generator = get_generator(args.generator, args.n_sample, args.seed)

    if args.compute_nfe:
        total, trials = 0.0, 300
        for _ in range(trials):
            x0 = generator.randn(1, 4, args.image_size // 8, args.image_size // 8).to(device)
            x0, kw = make_model_kwargs(args, x0, generator, device)
            _, nfe = sample_from_model(model, x0, kw, args)
            total += float(nfe) / trials
        print(f"Average NFE over {trials} trials: {int(total)}")
        return
    if args.measure_time:
        x = generator.randn(1, 4, args.image_size // 8, args.image_size // 8).to(device)
        for _ in range(10):
            model(torch.tensor(1.0, device=device), x)
        times = []
        for _ in range(300):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            run_sampling(model, vae, args, 1, generator, device)
            e.record()
            torch.cuda.synchronize()
            times.append(s.elapsed_time(e))
        tt = torch.tensor(times)
        print("Inference time: {:.2f}+/-{:.2f}ms".format(float(tt.mean()), float(tt.std(unbiased=False))))
        return
    save_dir = args.save_dir or "./generated_samples/{}/exp{}_ep{}_m{}".format(args.dataset, args.exp, args.epoch_id, args.method)
    if args.compute_fid:
        n = args.batch_size
        total_samples = int(math.ceil(args.n_sample / n) * n)
        t0 = time.time()
        if getattr(args, "fid_feature_extractor", ""):
            # on-device statistics (lfm_amd/fid.py): the images never leave the GPU; mu / sigma / Frechet distance as fid_score.py:230-283
            from .fid import FeatureStatistics, fid_against_reference_stats, load_feature_extractor

            extract = load_feature_extractor(args.fid_feature_extractor, device)
            stats = None
            for i in range(total_samples // n):
                feats = extract(images_to_uint8(run_sampling(model, vae, args, n, generator, device), rounding=True))
                stats = stats or FeatureStatistics(feats.shape[1], feats.device)
                stats.update(feats)
            fid = fid_against_reference_stats(stats.all_reduce(), args.real_img_dir)
            print("FID = {}".format(fid))
            with open(args.output_log or os.devnull, "a") as f:
                f.write("Epoch = {}, FID = {}\n".format(args.epoch_id, fid))
            return
        # batches in flight: consecutive batches alternate between HIP streams (solvers.concurrency_twin: the same weights, own scratch and captured graphs),
        # and the host converts / writes batch i - lanes while batches i - lanes + 1 .. i run (every lane stays busy under the JPEG encoding) -- the reference's
        # loop (:264-269) serialises solve, decode and JPEG writes
        from collections import deque

        lanes = make_lanes(model, vae, device, int(getattr(args, "in_flight", 0) or 0) or 2)
        pending = deque()

        def drain(p):
            p[0].synchronize()
            # the single-process script writes through torchvision.utils.save_image: ROUNDING uint8 conversion (:264-269)
            save_images_uint8(p[1], save_dir, p[2] * n)

        for i in range(total_samples // n):
            mdl, va, st = lanes[i % len(lanes)]
            with torch.cuda.stream(st):
                u8 = images_to_uint8(run_sampling(mdl, va, args, n, generator, device), rounding=True)
                done = torch.cuda.Event()
                done.record(st)
            pending.append((done, u8, i))
            if len(pending) > len(lanes):
                drain(pending.popleft())
        while pending:
            drain(pending.popleft())
        print(f"wrote {total_samples} images to {save_dir} in {time.time() - t0:.1f}s; FID needs pytorch_fid + Inception weights "
              "(not available offline): run the reference's pytorch_fid on that directory (lfm_amd.io_formats has the statistics "
              "reader and the Frechet distance)")
        return
    img = run_sampling(model, vae, args, args.batch_size, generator, device)
    # default mode: ONE sample sheet, nrow=8, padding=0, named as the reference names it (test_flow_latent.py:288-298)
    from .io_formats import make_grid_nhwc, save_jpeg

    if not args.use_karras_samplers:
        save_path = "./samples_{}_{}_{}_{}".format(args.dataset, args.method, args.atol, args.rtol)
    else:
        save_path = "./samples_{}_{}_{}".format(args.dataset, args.method, args.num_steps)
    if args.num_classes is not None and args.num_classes > 1:
        save_path += "_cfg{}".format(args.cfg_scale)
    save_path = os.path.join(args.save_dir, os.path.basename(save_path)) if args.save_dir else save_path
    save_path += ".jpg"
    if os.path.dirname(save_path):
        os.makedirs(os.path.dirname(save_path), exist_ok=True)
    save_jpeg(make_grid_nhwc(images_to_uint8(img, rounding=True).cpu(), nrow=8, padding=0), save_path)
    print("Samples are save at '{}".format(save_path))


if __name__ == "__main__":
    main()

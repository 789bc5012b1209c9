"""Wire / disk formats either side of the sampling path: checkpoints in, images and FID statistics out.

Host-side only; nothing here touches the GPU path.  Reference behaviour followed, with file:line:

* checkpoints   ``model_{epoch}.pth`` = flat ``state_dict`` of an accelerate/DDP-wrapped model, every key prefixed ``module.``
                (train_flow_latent.py:211-214, stripped unconditionally at test_flow_latent.py:140-141);
                ``content.pth`` = dict(epoch, global_step, args, model_dict, optimizer, scheduler) (train_flow_latent.py:196-203).
* images        single-process script: ``torchvision.utils.save_image`` = ``mul(255).add_(0.5).clamp_(0, 255).to(uint8)`` (ROUNDING),
                per-image JPEGs (test_flow_latent.py:269) or one nrow=8, padding=0 grid (:297);
                DDP script: ``clamp((x + 1) / 2, 0, 1) * 255 -> uint8`` (TRUNCATION) (test_flow_latent_ddp.py:131-139).
                Both end in ``PIL.Image.save(path)`` with PIL's default JPEG settings.
* FID           ``.npy`` / ``.npz`` statistics with the two layouts accepted at pytorch_fid/fid_score.py:254-260, activation statistics
                (:228-246) and the Frechet distance incl. its eps and imaginary-part rules (:177-225).  The Inception feature extractor
                itself needs torchvision + downloaded weights and is not available offline.
"""
import os

import numpy as np
import torch


# ------------------------------------------------------------------ checkpoints
def extract_state_dict(obj):
    """Flat, prefix-free state_dict from any checkpoint object the reference writes."""
    if isinstance(obj, dict) and "model_dict" in obj and isinstance(obj["model_dict"], dict):
        obj = obj["model_dict"]  # content.pth
    if not isinstance(obj, dict) or not obj or not all(isinstance(k, str) for k in obj):
        raise ValueError("not a state_dict / content.pth object")
    if not all(torch.is_tensor(v) for v in obj.values()):
        raise ValueError("checkpoint dict holds non-tensor values and no 'model_dict' entry")
    # the reference strips 7 characters from EVERY key; strip only a real 'module.' prefix
    return {(k[7:] if k.startswith("module.") else k): v for k, v in obj.items()}


def load_state_dict_file(path, map_location="cpu", trust_checkpoint=False):
    """Read ``model_{epoch}.pth`` / ``content.pth`` with the SAFE unpickler only.  ``content.pth`` pickles an ``argparse.Namespace``
    next to the tensors (train_flow_latent.py:196-203): that one class is allow-listed; anything else the safe loader rejects stays
    rejected (a file the safe loader refuses is exactly the file that must not reach the full unpickler), unless the caller passes
    ``trust_checkpoint=True`` explicitly (``--trust_checkpoint``), which is logged."""
    import argparse

    if trust_checkpoint:
        import warnings

        warnings.warn(f"loading {path} with the full (unsafe) unpickler because trust_checkpoint=True", stacklevel=2)
        return extract_state_dict(torch.load(path, map_location=map_location, weights_only=False))
    with torch.serialization.safe_globals([argparse.Namespace]):
        obj = torch.load(path, map_location=map_location, weights_only=True)
    return extract_state_dict(obj)


# ------------------------------------------------------------------ images
def to_uint8_rounding(img01):
    """torchvision.utils.save_image's conversion of a [0,1] float image tensor [...,C,H,W] -> uint8 [...,H,W,C]."""
    x = img01.detach().to("cpu", torch.float32)
    x = x.mul(255).add_(0.5).clamp_(0, 255)
    return x.movedim(-3, -1).to(torch.uint8)


def to_uint8_truncating(img_pm1):
    """test_flow_latent_ddp.py:131-135 on a [-1,1] image tensor [...,C,H,W] -> uint8 [...,H,W,C] (what lfm_images_to_uint8 does on the GPU)."""
    x = img_pm1.detach().to("cpu", torch.float32)
    x = torch.clamp((x + 1.0) / 2.0, 0, 1) * 255
    return x.movedim(-3, -1).to(torch.uint8)


def make_grid_nhwc(img_u8, nrow=8, padding=0):
    """Tile [N,H,W,C] uint8 images row-major, ``nrow`` per row, like torchvision.utils.make_grid(padding=0) does for the sample sheet."""
    if padding != 0:
        raise NotImplementedError("the reference only uses padding=0")
    n, h, w, c = img_u8.shape
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    grid = torch.zeros(rows * h, cols * w, c, dtype=torch.uint8)
    for k in range(n):
        r, q = divmod(k, cols)
        grid[r * h:(r + 1) * h, q * w:(q + 1) * w] = img_u8[k]
    return grid


def save_jpeg(hwc_u8, path):
    from PIL import Image

    a = hwc_u8.cpu().numpy() if torch.is_tensor(hwc_u8) else np.asarray(hwc_u8)
    Image.fromarray(a[..., 0] if a.shape[-1] == 1 else a).save(path)


def save_image_grid(img01, path, nrow=8, padding=0):
    """torchvision.utils.save_image(fake_image, path, padding=0, nrow=8) (test_flow_latent.py:297)."""
    save_jpeg(make_grid_nhwc(to_uint8_rounding(img01), nrow=nrow, padding=padding), path)


def save_indexed_jpegs(img_u8_nhwc, save_dir, start_index, world_size=1, rank=0):
    """One JPEG per image named by the reference's global index ``j * world + rank + total`` (test_flow_latent_ddp.py:138)."""
    os.makedirs(save_dir, exist_ok=True)
    for j in range(img_u8_nhwc.shape[0]):
        save_jpeg(img_u8_nhwc[j], os.path.join(save_dir, f"{j * world_size + rank + start_index}.jpg"))


# ------------------------------------------------------------------ FID statistics
def read_fid_stats(path):
    """(mu, sigma) from a ``.npz`` (keys mu, sigma) or a ``.npy`` holding a pickled dict (fid_score.py:254-260)."""
    f = np.load(path, allow_pickle=True)
    try:
        return f["mu"][:], f["sigma"][:]
    except (IndexError, KeyError, TypeError):
        d = f.item()
        return d["mu"][:], d["sigma"][:]


def activation_statistics(act):
    """fid_score.py:243-246: mean and (unbiased, rowvar=False) covariance of [n, dims] activations."""
    act = np.asarray(act, dtype=np.float64)
    return np.mean(act, axis=0), np.cov(act, rowvar=False)


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """fid_score.py:177-225, same order of operations (sqrtm of the product, eps retry, imaginary-part tolerance 1e-3)."""
    from scipy import linalg

    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    if mu1.shape != mu2.shape:
        raise ValueError("Training and test mean vectors have different lengths")
    if sigma1.shape != sigma2.shape:
        raise ValueError("Training and test covariances have different dimensions")
    diff = mu1 - mu2
    covmean = linalg.sqrtm(sigma1.dot(sigma2))
    if isinstance(covmean, tuple):  # older scipy returns (sqrtm, errest) with disp=False only; keep robust
        covmean = covmean[0]
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)

"""Model-constructor interface of the sampling path (drop-in for /root/reference/models/__init__.py:6-17)."""
from .DiT import DiT, DiT_models


def create_network(config):
    """``create_network(args) -> nn.Module`` with the reference's dispatch (models/__init__.py:6-17).

    DiT-* model types are built on the HIP path.  ``--use_origin_adm`` (guided-diffusion UNet) and the
    EDM ``adm`` are SURVEY.md §8 rows a14 / (f)1 and are not built yet: they raise instead of silently
    falling back to anything.
    """
    if getattr(config, "use_origin_adm", False):
        raise NotImplementedError("origin-ADM UNet (SURVEY.md §8 a14) is not built on the HIP path yet")
    if "DiT" not in config.model_type:
        raise NotImplementedError(f"model_type {config.model_type!r}: only DiT-* is built on the HIP path (SURVEY.md §8f)")
    return DiT_models[config.model_type](
        img_resolution=config.image_size // config.f,
        in_channels=config.num_in_channels,
        label_dropout=config.label_dropout,
        num_classes=config.num_classes,
    )


__all__ = ["create_network", "DiT", "DiT_models"]

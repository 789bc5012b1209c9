"""Model-constructor interface of the sampling path (drop-in for /root/reference/models/__init__.py:6-17)."""
from .DiT import DiT, DiT_models


def create_network(config):
    """``create_network(args) -> nn.Module`` with the reference's dispatch (models/__init__.py:6-17).

    DiT-*, ``--use_origin_adm`` (guided-diffusion ``UNetModel``) and the EDM ``adm`` (``DhariwalUNet``) are built
    on the HIP path.  ``ncsn++`` / ``ddpm++`` (SongUNet) and ``--layout`` (UNetModelAttn) are out of scope and raise instead of
    silently falling back to anything.
    """
    if getattr(config, "use_origin_adm", False):
        return get_flow_model(config)
    if "DiT" not in config.model_type:
        from .EDM import get_edm_network

        return get_edm_network(config)
    return DiT_models[config.model_type](
        img_resolution=config.image_size // config.f,
        in_channels=config.num_in_channels,
        label_dropout=config.label_dropout,
        num_classes=config.num_classes,
    )


def get_flow_model(config):
    """Origin-ADM constructor (reference models/__init__.py:20-70).  Flags the ``_ddp`` parser forgets
    (``use_scale_shift_norm`` etc., test_flow_latent_ddp.py:210-212) default to the single-file parser's values."""
    from .unet import UNetModel

    if getattr(config, "layout", False):
        raise NotImplementedError("--layout (UNetModelAttn with SpatialTransformer) is out of scope (SURVEY.md §2)")
    return UNetModel(
        image_size=config.image_size // 8,
        in_channels=config.num_in_channels,
        model_channels=config.nf,
        out_channels=config.num_out_channels,
        num_res_blocks=config.num_res_blocks,
        attention_resolutions=config.attn_resolutions,
        dropout=config.dropout,
        channel_mult=config.ch_mult,
        conv_resample=getattr(config, "resamp_with_conv", True),
        dims=2,
        num_classes=config.num_classes,
        use_checkpoint=False,
        use_fp16=False,
        num_heads=config.num_heads,
        num_head_channels=config.num_head_channels,
        num_heads_upsample=getattr(config, "num_head_upsample", -1),
        use_scale_shift_norm=getattr(config, "use_scale_shift_norm", True),
        resblock_updown=getattr(config, "resblock_updown", False),
        use_new_attention_order=getattr(config, "use_new_attention_order", False),
    )


__all__ = ["create_network", "get_flow_model", "DiT", "DiT_models"]

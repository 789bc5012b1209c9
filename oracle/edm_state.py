"""TEST INFRASTRUCTURE (not product code): a seeded, order-independent state maker for DhariwalUNet-shaped parameter trees
(/root/reference/models/EDM.py:716-861), so that full-size golden fixtures hold only inputs and the reference's outputs -- the 280-360 M parameters are
regenerated from (name, shape, seed) on both sides (oracle/make_golden.py::golden_edm_full with the unmodified reference module; tests/test_gpu_edm.py with the
product module, whose parameter names and shapes are the reference's).

Every tensor is drawn from its own generator (seed x crc32(name)), so the result does not depend on the order in which a module registers its parameters.
Values are fp16-representable (the product packs GEMM operands in fp16; the fixture is about the arithmetic, not about weight rounding).  The zero-initialised
layers of the reference (conv1 / proj / out_conv, EDM.py `init_zero`) get small non-zero weights -- every path of the network must matter -- sized so that the
residual stream stays O(1) over the ~40 blocks."""
import zlib

import torch


def seeded_edm_state(named_shapes, seed):
    """named_shapes: iterable of (name, shape) of the FLOATING tensors to fill (parameters; `resample_filter` buffers are left to the module).
    Returns {name: fp32 tensor}."""
    out = {}
    for name, shape in named_shapes:
        if name.endswith("resample_filter"):
            continue
        g = torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
        shape = tuple(shape)
        leaf = name.rsplit(".", 1)[-1]
        owner = name.rsplit(".", 2)[-2] if name.count(".") else ""
        if leaf == "weight" and len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            gain = 1.0
            if owner in ("conv1", "proj", "out_conv"):  # the reference's zero-initialised layers: small, so the residual stream does not blow up
                gain = 0.3
            elif owner == "map_label":  # init_weight = sqrt(label_dim) on kaiming_normal: unit-variance columns
                gain = 0.5 * fan_in ** 0.5
            v = torch.randn(shape, generator=g) * (gain / fan_in ** 0.5)
        elif leaf == "weight":  # GroupNorm scale
            v = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:  # biases (convolutions, linears, GroupNorm)
            v = 0.05 * torch.randn(shape, generator=g)
        out[name] = v.half().float()
    return out


def load_seeded(module, seed):
    """Fill `module` (reference or product DhariwalUNet) with the seeded state; returns the checksum the fixture records."""
    sd = module.state_dict()
    new = seeded_edm_state([(k, v.shape) for k, v in sd.items() if v is not None and v.is_floating_point()], seed)
    for k, v in new.items():
        sd[k] = v
    module.load_state_dict(sd, strict=True)
    return float(sum(v.double().abs().sum() for v in new.values()))


EDM_CONFIGS = {  # test_args/{ffhq,bed}_adm.txt + bash_scripts/run_test.sh:4-8,25-33;  imnet_adm.txt + run_test_cls.sh:13-17
    "ffhq_adm": dict(img_resolution=32, in_channels=4, out_channels=4, label_dim=0, augment_dim=0, model_channels=256, channel_mult=[1, 2, 3, 4],
                     channel_mult_emb=4, num_blocks=2, attn_resolutions=[16, 8, 4], dropout=0.0, label_dropout=0.0),
    "imnet_adm": dict(img_resolution=32, in_channels=4, out_channels=4, label_dim=1000, augment_dim=0, model_channels=256, channel_mult=[1, 2, 3, 4],
                      channel_mult_emb=4, num_blocks=2, attn_resolutions=[16, 8, 4], dropout=0.0, label_dropout=0.1),
}

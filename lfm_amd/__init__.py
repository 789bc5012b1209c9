"""lfm_amd -- MI355X-native sampling hot path of LFM (Flow Matching in Latent Space).

Host side (Python, mirrors the reference's own module layout for this path):
    lfm_amd.models.create_network      <- /root/reference/models/__init__.py:6-17
    lfm_amd.models.DiT                 <- /root/reference/models/DiT.py
    lfm_amd.sampler.*                  <- /root/reference/sampler/{karras_sample,random_util}.py
    lfm_amd.solvers                    <- torchdiffeq as called at test_flow_latent.py:42-76
    lfm_amd.autoencoder.AutoencoderKL  <- diffusers AutoencoderKL (decode only)
Device side: lfm_amd/csrc/*.hip -> lfm_amd/_lib/liblfm_hip.so (C ABI in include/lfm_hip.h).
There is NO fallback path: a forward on a device without the HIP library raises.
"""
__version__ = "0.1.0"

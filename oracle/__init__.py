"""CPU oracle for the LFM sampling hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker.  The shipped path lives in ``lfm_amd/`` and
fails loudly when the HIP library is missing; it never routes through here.

Pinning status (see DESIGN.md "Oracle"):

* ``dit_ref`` / ``unet_ref`` / ``karras_ref`` / ``randgen_ref`` are PINNED: they
  are checked in ``tests/test_oracle_golden.py`` against golden vectors that
  ``oracle/make_golden.py`` produced by importing the unmodified reference
  classes from ``/root/reference`` (behind the three-class ``timm`` shim in
  ``oracle/timm_shim.py``).
* The EDM ``DhariwalUNet`` and the FID back half have no separate restatement:
  the goldens ``edm_tiny.pt`` (unmodified ``models/EDM.py``) and ``fid.pt``
  (``pytorch_fid/fid_score.py::calculate_frechet_distance``) pin the product's
  ``lfm_amd/models/EDM.py`` and ``lfm_amd/io_formats.py`` directly.
* ``ode_ref`` (torchdiffeq) and ``vae_ref`` (diffusers ``AutoencoderKL``
  decoder) restate third-party packages that are neither vendored in the
  reference nor installed here (requirements.txt:2-3, no versions):
  **parity unpinned** against those packages themselves.  Independent anchors:
  scipy's RK45 / RK23 / DOP853 and the Runge-Kutta order conditions for
  ``ode_ref`` (tests/test_ode_ref.py); for ``vae_ref`` the ``transformers``
  package's own port of the latent-diffusion Encoder / Decoder (the network
  ``AutoencoderKL`` ports), which reproduces ``vae_decode`` /
  ``vae_encode_moments`` from our state dict (tests/test_vae_ref_ldm.py), the
  published parameter counts and the frozen diffusers key list
  (tests/test_vae_ref.py); ``torch.nn.MultiheadAttention`` / ``F.unfold`` for
  the timm shim (tests/test_timm_shim.py).
"""

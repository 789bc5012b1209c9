"""Batch-size / world-size independent noise (drop-in for /root/reference/sampler/random_util.py).

``get_generator(kind, num_samples, seed)`` -> object with ``randn / randint / randn_like``.

* ``determ`` / ``determ-indiv`` reproduce the reference bit-for-bit on CPU (tests/golden/randgen.pt): the
  reference draws the WHOLE ``(num_samples, ...)`` tensor from one ``torch.Generator`` and returns rows
  ``done + rank + k*world`` (random_util.py:58-75), so the same rows come out for any batch size or
  world size.  That costs an 819 MB CPU draw per call at 50k samples; it is kept because bit-parity of the
  latents with the reference is the point of this class.
* ``device`` (new): a per-rank ``torch.Generator`` on the accelerator that draws only the requested rows,
  directly in HBM -- what the benchmark and the production sampler use.
"""
import torch as th
import torch.distributed as dist


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def get_generator(generator, num_samples=0, seed=0, device=None):
    if generator == "dummy":
        return DummyGenerator()
    if generator == "determ":
        return DeterministicGenerator(num_samples, seed)
    if generator == "determ-indiv":
        return DeterministicIndividualGenerator(num_samples, seed)
    if generator == "device":
        return DeviceGenerator(seed, device)
    raise NotImplementedError(generator)


class DummyGenerator:
    def randn(self, *args, **kwargs):
        return th.randn(*args, **kwargs)

    def randint(self, *args, **kwargs):
        return th.randint(*args, **kwargs)

    def randn_like(self, *args, **kwargs):
        return th.randn_like(*args, **kwargs)


class _Deterministic:
    def __init__(self, num_samples, seed):
        self.rank, self.world_size = _rank_world()
        self.num_samples = num_samples
        self.done_samples = 0
        self.seed = seed

    def _indices(self, n):
        idx = th.arange(self.done_samples + self.rank, self.done_samples + self.world_size * int(n), self.world_size)
        idx = th.clamp(idx, 0, self.num_samples - 1)
        assert len(idx) == n, f"rank={self.rank}, ws={self.world_size}, l={len(idx)}, bs={n}"
        return idx

    def randn_like(self, tensor):
        return self.randn(*tensor.size(), dtype=tensor.dtype, device=tensor.device)

    def get_seed(self):
        return self.seed


class DeterministicGenerator(_Deterministic):
    """One generator per device type; every call draws (num_samples, *size[1:]) and sub-samples rows."""

    def __init__(self, num_samples, seed=0):
        super().__init__(num_samples, seed)
        self.rng_cpu = th.Generator()
        self.rng_cuda = th.Generator("cuda") if th.cuda.is_available() else None
        self.set_seed(seed)

    def get_generator(self, device):
        if th.device(device).type == "cpu":
            return self.rng_cpu
        if self.rng_cuda is None:
            raise RuntimeError("no accelerator generator available")
        return self.rng_cuda

    def randn(self, *size, dtype=th.float, device="cpu"):
        idx = self._indices(size[0])
        return th.randn(self.num_samples, *size[1:], generator=self.get_generator(device), dtype=dtype, device=device)[idx.to(device)]

    def randint(self, low, high, size, dtype=th.long, device="cpu"):
        idx = self._indices(size[0])
        return th.randint(low, high, generator=self.get_generator(device), size=(self.num_samples, *size[1:]), dtype=dtype,
                          device=device)[idx.to(device)]

    def set_done_samples(self, done_samples):
        self.done_samples = done_samples
        self.set_seed(self.seed)

    def set_seed(self, seed):
        self.rng_cpu.manual_seed(seed)
        if self.rng_cuda is not None:
            self.rng_cuda.manual_seed(seed)


class DeterministicIndividualGenerator(_Deterministic):
    """One generator per sample (seeded i + num_samples*seed): less memory, same independence of batch/world size."""

    def __init__(self, num_samples, seed=0):
        super().__init__(num_samples, seed)
        self.rng_cpu = [th.Generator() for _ in range(num_samples)]
        self.rng_cuda = [th.Generator("cuda") for _ in range(num_samples)] if th.cuda.is_available() else None
        self.set_seed(seed)

    def get_generator(self, device):
        return self.rng_cpu if th.device(device).type == "cpu" else self.rng_cuda

    def randn(self, *size, dtype=th.float, device="cpu"):
        gens = self.get_generator(device)
        return th.cat([th.randn(1, *size[1:], generator=gens[i], dtype=dtype, device=device) for i in self._indices(size[0])], 0)

    def randint(self, low, high, size, dtype=th.long, device="cpu"):
        gens = self.get_generator(device)
        return th.cat([th.randint(low, high, generator=gens[i], size=(1, *size[1:]), dtype=dtype, device=device)
                       for i in self._indices(size[0])], 0)

    def set_done_samples(self, done_samples):
        self.done_samples = done_samples

    def set_seed(self, seed):
        for i, g in enumerate(self.rng_cpu):
            g.manual_seed(i + self.num_samples * seed)
        if self.rng_cuda is not None:
            for i, g in enumerate(self.rng_cuda):
                g.manual_seed(i + self.num_samples * seed)


class DeviceGenerator:
    """Draws only the requested rows, on the accelerator, from a rank-offset seed (seed + rank as the reference's
    drivers do, test_flow_latent_ddp.py:30).  Not bit-compatible with ``determ`` -- statistically equivalent."""

    def __init__(self, seed=0, device=None):
        self.rank, self.world_size = _rank_world()
        self.device = th.device(device if device is not None else ("cuda" if th.cuda.is_available() else "cpu"))
        self.gen = th.Generator(self.device)
        self.gen.manual_seed(seed + self.rank)

    def randn(self, *size, dtype=th.float, device=None):
        return th.randn(*size, generator=self.gen, dtype=dtype, device=self.device)

    def randint(self, low, high, size, dtype=th.long, device=None):
        return th.randint(low, high, size, generator=self.gen, dtype=dtype, device=self.device)

    def randn_like(self, tensor):
        return self.randn(*tensor.size(), dtype=tensor.dtype)

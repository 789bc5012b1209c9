from .karras_sample import karras_sample, sample_euler, sample_heun  # noqa
from .random_util import get_generator  # noqa

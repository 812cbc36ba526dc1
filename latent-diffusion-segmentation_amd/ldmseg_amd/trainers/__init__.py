from .sampler import TrainerDiffusion  # noqa: F401

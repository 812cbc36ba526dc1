from .unet import UNet  # noqa: F401
from .vae import GeneralVAESeg, DiagonalGaussianDistribution  # noqa: F401

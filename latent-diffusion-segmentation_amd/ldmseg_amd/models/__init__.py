from .unet import UNet  # noqa: F401
from .vae import GeneralVAESeg, GeneralVAEImage, DiagonalGaussianDistribution  # noqa: F401

"""MI355X-native U-Net / U-Net-DC forward+backward path (gfx950 HIP kernels behind the
reference's nn.Module surface).  See DESIGN.md."""
from .unet import UNet, UNetDC  # noqa: F401

__all__ = ["UNet", "UNetDC"]
__version__ = "0.1.0"

"""Drop-in for /root/reference/models/model.py (plain U-Net, all dilations 1)."""
from unet_dc_segmentation_amd.unet import UNet  # noqa: F401

__all__ = ["UNet"]

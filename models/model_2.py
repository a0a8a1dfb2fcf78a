"""Drop-in for /root/reference/models/model_2.py: ``from models.model_2 import UNetDC``
(train_DC_focal.py:18, quantify_droplets_batch.py:26) resolves to the MI355X-native module."""
from unet_dc_segmentation_amd.unet import UNetDC  # noqa: F401

__all__ = ["UNetDC"]

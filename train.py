#!/usr/bin/env python3
"""Drop-in for the reference's ``train.py``: the plain U-Net (all dilations 1, models/model.py)
trained with BCE+Dice for 50 epochs (train.py:123-126).  Same loop as train_DC_focal.py."""
from train_DC_focal import build_parser, main

if __name__ == "__main__":
    main(parser=build_parser(arch="unet", epochs=50, ckpt="best_UNet_model.pth", loss="bce_dice"))

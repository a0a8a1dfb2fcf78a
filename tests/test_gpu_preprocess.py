"""GPU: input preprocessing kernels (csrc/preprocess.hip) against the numpy restatement of the OpenCV operators in
utils/data_loader.py (the CPU path of the entry points).  Byte arithmetic: the bar is bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _image(rng, h, w, c):
    img = (rng.random((h, w, c)) * 70).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    img += (40 * np.sin(yy / 37.0)[..., None] + 40).astype(np.uint8)          # smooth background the opening must follow
    for _ in range(60):
        cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(2, 14)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = rng.integers(180, 255)
    return img


@pytest.mark.parametrize("case", [(96, 130, 3, 15), (257, 301, 3, 50), (64, 64, 1, 1), (200, 90, 3, 2), (150, 150, 4, 127),
                                  (1040, 1388, 3, 50), (7, 5, 3, 3), (40, 33, 2, 9), (130, 70, 1, 128)])
def test_rolling_ball_matches_numpy_restatement(case):
    from unet_dc_segmentation_amd.preprocess import rolling_ball_device
    from utils.data_loader import rolling_ball_correction_rgb
    h, w, c, k = case
    img = _image(np.random.default_rng(h + k), h, w, c)
    ref = rolling_ball_correction_rgb(img, k)
    out = rolling_ball_device(torch.from_numpy(img).cuda(), k).cpu().numpy()
    assert out.dtype == np.uint8 and np.array_equal(out, ref)
    assert k == 1 or (int(ref.max()) == 255 and int(ref.min()) == 0)      # a 1 x 1 element: background = image, all zeros


def test_rolling_ball_flat_image_gives_zeros():
    """max == min after the subtraction: cv2.normalize's scale is 0, the result all zeros."""
    from unet_dc_segmentation_amd.preprocess import rolling_ball_device
    img = np.full((70, 45, 3), 93, np.uint8)
    out = rolling_ball_device(torch.from_numpy(img).cuda(), 9).cpu().numpy()
    assert not out.any()


@pytest.mark.parametrize("shape", [(1040, 1388), (512, 512), (276, 408), (2048, 100), (33, 17)])
def test_resize_to_network_input_matches_numpy_restatement(shape):
    from unet_dc_segmentation_amd.preprocess import resize_to_input_device
    from utils.data_loader import resize_linear_cv2_u8
    h, w = shape
    img = _image(np.random.default_rng(w), h, w, 3)
    ref = resize_linear_cv2_u8(img, 512, 512).astype(np.float32) / 255.0
    out = resize_to_input_device(torch.from_numpy(img).cuda(), 512).cpu().numpy()
    assert out.shape == (3, 512, 512) and out.dtype == np.float32
    assert np.array_equal(out, ref.transpose(2, 0, 1))

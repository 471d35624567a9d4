"""Image-space helpers with the reference's names and contracts (utils/proc.py), backed by the HIP
kernels in libisx_hip.so.  Inputs may be numpy arrays or DeviceImage; results that are images are
DeviceImage (np.asarray() downloads them).  There is no CPU path: every function needs a HIP device.

  get_corners_from_image   reference utils/proc.py:34-45
  seamless_merge           reference utils/proc.py:48-82      (alpha blender)
  seamless_gradient_merge  reference utils/proc.py:192-254    (feather blender; mutates img2)
  pad_image / get_padded_images   :280-304
  crop_image               :307-312
  transform_keypoints_from_roi    :315-322
  get_roi_from_corners     :325-333
  resize_image / load_image       :12-31   (stay on cv2, as in the reference)
"""
import numpy as np
import torch

from .. import ops
from ..device_image import DeviceImage


def _device(device=None):
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def _dev_img(img, device=None):
    return DeviceImage.from_any(img, _device(device))


def resize_image(image, to_resize=None):
    if to_resize is not None:
        h, w = to_resize
        if h == len(image) and w == len(image[0]):
            return image
        import cv2  # stays on cv2 (reference utils/proc.py:12-25, including its (h, w) -> dsize quirk)
        area_new, area_old = h * w, image.shape[0] * image.shape[1]
        interp = cv2.INTER_AREA if area_new < area_old else (cv2.INTER_CUBIC if area_new > area_old else cv2.INTER_LINEAR)
        return cv2.resize(np.asarray(image), (h, w), interpolation=interp)
    return image


def load_image(filepath, to_resize=None):
    """cv2.imread(filepath) (BGR, 3 channels, alpha dropped; None if unreadable) when cv2 is importable, else the
    same array through PIL -- the counterpart of async_save.write_image."""
    try:
        import cv2
        image = cv2.imread(filepath)
    except ImportError:
        from PIL import Image
        try:
            with Image.open(filepath) as im:
                image = np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])
        except (OSError, ValueError):
            image = None  # cv2.imread's failure value
    return resize_image(image, to_resize)


def get_corners_from_image(image, device=None):
    d = _dev_img(image, device)
    return ops.bbox_to_corners(ops.bbox_nonzero(d.t).cpu().numpy())


def get_roi_from_corners(image, p1, p2, img=True):
    x, y = p1
    w, h = p2[0] - p1[0], p2[1] - p1[1]
    if img is True:
        return image[y:y + h, x:x + w]
    return (y, y + h, x, x + w)


def crop_image(im, device=None):
    d = _dev_img(im, device)
    corners = ops.bbox_to_corners(ops.bbox_nonzero(d.t).cpu().numpy())
    if corners is False:
        return False
    return get_roi_from_corners(d, corners[0], corners[1])


def transform_keypoints_from_roi(keypoints_roi, roi_topleft):
    x, y = roi_topleft
    for kp in keypoints_roi:
        kp.pt = (kp.pt[0] + x, kp.pt[1] + y)
    return list(keypoints_roi)


def pad_image(new_image_height, new_image_width, old_image_width, old_image_height, img, channels=3, device=None):
    d = _dev_img(img, device)
    if channels not in (1, 3):
        raise ValueError("pad_image supports 1 or 3 channels (the reference's fill tuple has 3 entries)")
    shape = (new_image_height, new_image_width) if channels == 1 else (new_image_height, new_image_width, channels)
    out = torch.zeros(shape, dtype=torch.uint8, device=d.t.device)
    xc = (new_image_width - old_image_width) // 2
    yc = (new_image_height - old_image_height) // 2
    ops.copy2d(d.t, out[yc:yc + old_image_height, xc:xc + old_image_width])
    return DeviceImage(out)


def get_padded_images(img, ref, device=None):
    new_w = ref.shape[1] + 2 * img.shape[1]
    new_h = ref.shape[0] + 2 * img.shape[0]
    channels = 1 if len(img.shape) == 2 else img.shape[2]
    new_ref = pad_image(new_h, new_w, ref.shape[1], ref.shape[0], ref, channels=channels, device=device)
    new_img = pad_image(new_h, new_w, img.shape[1], img.shape[0], img, channels=channels, device=device)
    return new_img, new_ref


def seamless_merge(image1, image2, ref_image_contrib, device=None):
    a, b = _dev_img(image1, device), _dev_img(image2, device)
    return DeviceImage(ops.alpha_blend(a.t, b.t, ref_image_contrib))


def seamless_gradient_merge(warped_img1, img2, feather_pixels=10, device=None):
    """Mutates and returns img2 when it is a DeviceImage (as the reference mutates its ndarray)."""
    a, b = _dev_img(warped_img1, device), _dev_img(img2, device)
    corners = ops.bbox_to_corners(ops.bbox_nonzero(a.t).cpu().numpy())
    if corners is False:
        return b
    (x0, y0), (x1, y1) = corners
    x0, y0 = max(0, x0 - 50), max(0, y0 - 50)
    x1, y1 = min(b.shape[1], x1 + 50), min(b.shape[0], y1 + 50)
    ops.gradient_blend_roi(a.t[y0:y1, x0:x1], b.t[y0:y1, x0:x1], feather_pixels)
    b.invalidate()
    return b

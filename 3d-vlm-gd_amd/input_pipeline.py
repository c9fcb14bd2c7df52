"""Input pipeline on device — host-side mirror of

* `vggt/utils/load_fn.py:12-146` `load_and_preprocess_images(image_path_list, mode)`: same name, arguments, errors and
  result; the file decode (PIL open, alpha composite onto white, convert("RGB"): load_fn.py:54-66) stays on the host,
  everything after it — PIL's bicubic resize on uint8, ToTensor, centre crop, white padding, batching — runs on the GPU
  and is bit-exact against the reference function (fixture G20);
* `data_utils/dataset_mast3r_scannetpp.py:185-207` `AugmentedCustomScanNetPPDataset`: ColorJitter(0.2, 0.2, 0.2, 0.1) +
  GaussianBlur(blur_limit=(3, 7)), each applied with albumentations' default p = 0.5, on uint8 images (`ColorAug`,
  `augment_sample`).  albumentations / OpenCV are not available: restated, parity unpinned (see csrc/image_prep.hip).

No CPU fallback: a missing library raises GdHipError."""
import ctypes
import random

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream

TARGET_SIZE = 518


def _coeffs(in_size, out_size, device):
    """Pillow's resampling window / fixed-point coefficients (host, C ABI) -> device int32 tensors."""
    L = lib()
    ks = L.gd_pil_resample_ksize(in_size, out_size)
    if ks < 0:
        raise _lib.GdHipError(L.gd_last_error().decode())
    xmin, cnt, kk = np.empty(out_size, np.int32), np.empty(out_size, np.int32), np.empty(out_size * ks, np.int32)
    check(L.gd_pil_resample_coeffs(in_size, out_size, xmin.ctypes.data, cnt.ctypes.data, kk.ctypes.data), "gd_pil_resample_coeffs")
    to = lambda a: torch.from_numpy(a).to(device)
    return to(xmin), to(cnt), to(kk), ks


def pil_resize_bicubic(img_u8, new_w, new_h):
    """uint8 [H, W, C] CUDA tensor -> uint8 [new_h, new_w, C]: PIL `Image.resize((new_w, new_h), Image.Resampling.BICUBIC)`."""
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.is_cuda
    img_u8 = img_u8.contiguous()
    H, W, C = img_u8.shape
    dev = img_u8.device
    xh, ch, kh, ksh = _coeffs(W, new_w, dev) if new_w != W else (None, None, None, 0)
    xv, cv, kv, ksv = _coeffs(H, new_h, dev) if new_h != H else (None, None, None, 0)
    tmp = torch.empty(H, new_w, C, dtype=torch.uint8, device=dev) if (new_w != W and new_h != H) else None
    out = torch.empty(new_h, new_w, C, dtype=torch.uint8, device=dev)
    check(lib().gd_pil_resize_bicubic_u8(ptr(img_u8), ptr(tmp), ptr(out), H, W, C, new_h, new_w, ptr(xh), ptr(ch), ptr(kh), ksh,
                                         ptr(xv), ptr(cv), ptr(kv), ksv, stream()), "gd_pil_resize_bicubic_u8")
    return out


def _geometry(width, height, mode):
    """load_fn.py:70-84"""
    if mode == "pad":
        if width >= height:
            new_width = TARGET_SIZE
            new_height = round(height * (new_width / width) / 14) * 14
        else:
            new_height = TARGET_SIZE
            new_width = round(width * (new_height / height) / 14) * 14
    else:
        new_width = TARGET_SIZE
        new_height = round(height * (new_width / width) / 14) * 14
    return new_width, new_height


def preprocess_images(images_u8, mode="crop", device="cuda"):
    """Decoded uint8 RGB images (numpy / tensors [H, W, 3]) -> float32 [N, 3, H', W'] on `device` (load_fn.py:67-146)."""
    if len(images_u8) == 0:
        raise ValueError("At least 1 image is required")
    if mode not in ["crop", "pad"]:
        raise ValueError("Mode must be either 'crop' or 'pad'")
    resized, shapes = [], []
    for im in images_u8:
        t = (torch.from_numpy(np.ascontiguousarray(im)) if isinstance(im, np.ndarray) else im).to(device)
        h, w = t.shape[:2]
        nw, nh = _geometry(w, h, mode)
        r = pil_resize_bicubic(t, nw, nh)
        crop_y0, crop_h = ((nh - TARGET_SIZE) // 2, TARGET_SIZE) if (mode == "crop" and nh > TARGET_SIZE) else (0, nh)
        ch, cw = (TARGET_SIZE, TARGET_SIZE) if mode == "pad" else (crop_h, nw)       # shape after the per-image crop / pad
        resized.append((r, crop_y0, crop_h, ch, cw))
        shapes.append((ch, cw))
    if len(set(shapes)) > 1:
        print(f"Warning: Found images with different shapes: {set(shapes)}")
    Hc, Wc = max(s[0] for s in shapes), max(s[1] for s in shapes)
    out = torch.empty(len(resized), 3, Hc, Wc, dtype=torch.float32, device=device)
    for i, (r, crop_y0, crop_h, ch, cw) in enumerate(resized):
        nh, nw = r.shape[:2]
        # two nested centred paddings (per-image square, then the batch's common shape): both put the floor half first
        top = (ch - crop_h) // 2 + (Hc - ch) // 2
        left = (cw - nw) // 2 + (Wc - cw) // 2
        check(lib().gd_u8_to_chw_float(ptr(r), ptr(out[i]), nh, nw, 3, crop_y0, crop_h, top, left, Hc, Wc, 1.0, stream()),
              "gd_u8_to_chw_float")
    return out


def load_and_preprocess_images(image_path_list, mode="crop", device="cuda"):
    """vggt/utils/load_fn.py:12-146 — same signature (+ `device`); decode on the host, the rest on the GPU."""
    from PIL import Image
    if len(image_path_list) == 0:
        raise ValueError("At least 1 image is required")
    if mode not in ["crop", "pad"]:
        raise ValueError("Mode must be either 'crop' or 'pad'")
    decoded = []
    for path in image_path_list:
        img = Image.open(path)
        if img.mode == "RGBA":
            img = Image.alpha_composite(Image.new("RGBA", img.size, (255, 255, 255, 255)), img)
        decoded.append(np.asarray(img.convert("RGB")).copy())
    return preprocess_images(decoded, mode=mode, device=device)


class ColorAug:
    """`A.Compose([A.ColorJitter(0.2, 0.2, 0.2, 0.1), A.GaussianBlur(blur_limit=(3, 7))])`
    (data_utils/dataset_mast3r_scannetpp.py:189-192) on a batch of uint8 RGB images [n, H, W, 3] on the GPU."""

    def __init__(self, brightness=0.2, contrast=0.2, saturation=0.2, hue=0.1, blur_limit=(3, 7), p_jitter=0.5, p_blur=0.5, seed=None):
        self.ranges = [(1 - brightness, 1 + brightness), (1 - contrast, 1 + contrast), (1 - saturation, 1 + saturation), (-hue, hue)]
        self.blur_limit, self.p_jitter, self.p_blur = blur_limit, p_jitter, p_blur
        self.rng = random.Random(seed)

    def sample(self, n):
        """-> (factors [n,4] float32, order [n,4] int32 (-1 = skipped), ksize [n] int32 (0 = no blur))."""
        f, o, k = np.zeros((n, 4), np.float32), np.full((n, 4), -1, np.int32), np.zeros(n, np.int32)
        for i in range(n):
            if self.rng.random() < self.p_jitter:
                f[i] = [self.rng.uniform(a, b) for a, b in self.ranges]
                order = [0, 1, 2, 3]
                self.rng.shuffle(order)
                o[i] = order
            if self.rng.random() < self.p_blur:
                k[i] = self.rng.randrange(self.blur_limit[0], self.blur_limit[1] + 1, 2)
        return f, o, k

    def apply(self, imgs_u8, factors, order, ksize):
        assert imgs_u8.dtype == torch.uint8 and imgs_u8.dim() == 4 and imgs_u8.shape[-1] == 3 and imgs_u8.is_cuda
        imgs_u8 = imgs_u8.contiguous()
        n, H, W, _ = imgs_u8.shape
        dev = imgs_u8.device
        f = torch.as_tensor(factors, dtype=torch.float32, device=dev).contiguous()
        o = torch.as_tensor(order, dtype=torch.int32, device=dev).contiguous()
        k = torch.as_tensor(ksize, dtype=torch.int32, device=dev).contiguous()
        ws = torch.empty(n, dtype=torch.int64, device=dev)
        jit = torch.empty_like(imgs_u8)
        check(lib().gd_color_jitter_u8(ptr(imgs_u8), ptr(jit), n, H, W, ptr(f), ptr(o), ptr(ws), stream()), "gd_color_jitter_u8")
        tmp = torch.empty(n, H, W, 3, dtype=torch.float32, device=dev)
        out = torch.empty_like(imgs_u8)
        check(lib().gd_gaussian_blur_u8(ptr(jit), ptr(tmp), ptr(out), n, H, W, ptr(k), stream()), "gd_gaussian_blur_u8")
        return out

    def __call__(self, imgs_u8):
        return self.apply(imgs_u8, *self.sample(imgs_u8.shape[0]))


def augment_sample(sample, aug, augmentation=True):
    """AugmentedCustomScanNetPPDataset.__getitem__ (data_utils/dataset_mast3r_scannetpp.py:199-209) on device tensors:
    sample['rgb_1'], sample['rgb_2'] float [3, H, W] in [0, 1] -> * 255 -> uint8 (truncation) -> augment -> / 255 float32."""
    for idx in (1, 2):
        x = sample[f"rgb_{idx}"]
        if augmentation:
            u8 = (x.permute(1, 2, 0) * 255).to(torch.uint8)[None]
            x = aug(u8)[0].permute(2, 0, 1) / 255.0
        sample[f"rgb_{idx}"] = x.to(torch.float32)
    return sample

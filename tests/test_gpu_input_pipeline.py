"""GPU parity of the input pipeline: `load_and_preprocess_images` against fixture G20 (written by the
reference's own function: bit-exact), the PIL resampler against the oracle on more shapes, the colour augmentation against
its restatement (parity unpinned: albumentations / OpenCV are absent)."""
import numpy as np
import pytest
import torch

import gd_oracle as O
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["crop", "pad"])
def test_preprocess_images_g20_bit_exact(mode):
    from gd_amd.input_pipeline import preprocess_images
    g = load_golden("g20_load_fn")
    ins = [g[f"{mode}.in{i}"].numpy() for i in range(2)]
    out = preprocess_images(ins, mode=mode)
    assert out.dtype == torch.float32 and out.is_cuda
    assert torch.equal(out.cpu(), g[f"{mode}.out_u8"].float() / 255)


@pytest.mark.parametrize("h,w,nh,nw", [(37, 53, 80, 53), (90, 120, 90, 64), (64, 48, 140, 210), (300, 200, 98, 70), (33, 31, 33, 31)])
def test_pil_bicubic_resize_matches_oracle(h, w, nh, nw):
    from gd_amd.input_pipeline import pil_resize_bicubic
    a = np.random.default_rng(h * w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    a[::7] = 255
    a[:, ::5] = 0                                     # hard edges: the cubic's negative lobes clip at both ends
    got = pil_resize_bicubic(torch.from_numpy(a).cuda(), nw, nh).cpu().numpy()
    assert np.array_equal(got, O.pil_resize_bicubic_u8(a, nw, nh))


def test_load_and_preprocess_images_files_and_errors(tmp_path):
    from PIL import Image
    from gd_amd.input_pipeline import load_and_preprocess_images
    rng = np.random.default_rng(3)
    paths, decoded = [], []
    for i, (h, w, c) in enumerate([(50, 70, 3), (64, 40, 4)]):
        a = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        p = str(tmp_path / f"im{i}.png")
        Image.fromarray(a, "RGBA" if c == 4 else "RGB").save(p)
        paths.append(p)
        im = Image.open(p)
        if im.mode == "RGBA":
            im = Image.alpha_composite(Image.new("RGBA", im.size, (255, 255, 255, 255)), im)
        decoded.append(np.asarray(im.convert("RGB")).copy())
    for mode in ("crop", "pad"):
        out = load_and_preprocess_images(paths, mode=mode)
        assert torch.equal(out.cpu(), O.preprocess_images(decoded, mode=mode))
        assert out.shape[-1] == 518 and out.shape[-2] % 14 == 0
    one = load_and_preprocess_images(paths[:1])
    assert one.dim() == 4 and one.shape[0] == 1
    with pytest.raises(ValueError):
        load_and_preprocess_images([])
    with pytest.raises(ValueError):
        load_and_preprocess_images(paths, mode="stretch")


def test_color_aug_matches_restatement():
    from gd_amd.input_pipeline import ColorAug, augment_sample
    rng = np.random.default_rng(5)
    n, H, W = 6, 45, 61
    imgs = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    imgs[0, :10] = 255
    imgs[1, :, :10] = 0
    aug = ColorAug(seed=11, p_jitter=0.8, p_blur=0.6)
    f, o, k = aug.sample(n)
    o[2] = [3, 1, -1, 0]          # a skipped operation
    k[3], k[4] = 7, 0
    out = aug.apply(torch.from_numpy(imgs).cuda(), f, o, k).cpu().numpy()
    bad = 0
    for i in range(n):
        ref = O.gaussian_blur_u8(O.color_jitter_u8(imgs[i], f[i], o[i]), int(k[i]))
        d = np.abs(out[i].astype(int) - ref.astype(int))
        assert d.max() <= 1, (i, d.max())          # float32 on both sides; exp / division may round one LSB apart
        bad += int((d > 0).sum())
    assert bad <= 1e-3 * imgs.size, bad
    # identity parameters leave the image alone up to the uint8 HSV round trip; no-op samples are exact copies
    same = aug.apply(torch.from_numpy(imgs).cuda(), np.zeros((n, 4), np.float32), np.full((n, 4), -1, np.int32), np.zeros(n, np.int32))
    assert torch.equal(same.cpu(), torch.from_numpy(imgs))
    s = {"rgb_1": torch.rand(3, 32, 40).cuda(), "rgb_2": torch.rand(3, 32, 40).cuda()}
    s2 = augment_sample(dict(s), ColorAug(seed=1))
    assert s2["rgb_1"].shape == (3, 32, 40) and s2["rgb_1"].dtype == torch.float32 and 0 <= s2["rgb_1"].min() and s2["rgb_1"].max() <= 1

"""G20: the reference's `load_and_preprocess_images` (vggt/utils/load_fn.py:12-146) run on synthetic
image files — build container only.  Writes tests/golden/g20_load_fn.npz: the decoded uint8 inputs and the function's
outputs (stored as uint8 = output * 255, exact: ToTensor is uint8 / 255) for crop and pad mode, incl. a down-scaled image
(PIL's antialiasing support), a centre-cropped tall image, an RGBA image and the white padding to a common shape.
Asserts that oracle/gd_oracle.py `preprocess_images` reproduces the outputs bit for bit."""
import os
import sys
import tempfile

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import ref_import as R  # noqa: E402
import gd_oracle as O  # noqa: E402

R.install()
from vggt.utils.load_fn import load_and_preprocess_images  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
rng = np.random.default_rng(20)


def synth(h, w, c=3):
    """smooth content + a little noise + hard edges (so that every branch of the cubic and the clip to [0, 255] fires)."""
    low = rng.integers(0, 256, (max(h // 16, 2), max(w // 16, 2), c), dtype=np.uint8)
    img = np.asarray(Image.fromarray(low if c > 1 else low[:, :, 0]).resize((w, h), Image.Resampling.BILINEAR)).reshape(h, w, c).astype(np.int32)
    img[: h // 4, : w // 4] += rng.integers(-6, 7, img[: h // 4, : w // 4].shape)     # noise on one corner (keeps the fixture small)
    img[h // 3:h // 3 + 4, :, :] = 255
    img[:, w // 2:w // 2 + 3, :] = 0
    return np.clip(img, 0, 255).astype(np.uint8)


cases = {"crop": [synth(120, 160), synth(200, 160)], "pad": [synth(420, 660), synth(90, 60, 4)]}
arrs = {}
with tempfile.TemporaryDirectory() as d:
    for mode, imgs in cases.items():
        paths, decoded = [], []
        for i, a in enumerate(imgs):
            p = os.path.join(d, f"{mode}{i}.png")
            Image.fromarray(a, "RGBA" if a.shape[2] == 4 else "RGB").save(p)
            paths.append(p)
            im = Image.open(p)
            if im.mode == "RGBA":      # the decode step of load_fn.py:57-64 (host side in the build as well)
                im = Image.alpha_composite(Image.new("RGBA", im.size, (255, 255, 255, 255)), im)
            decoded.append(np.asarray(im.convert("RGB")).copy())
        ref = load_and_preprocess_images(paths, mode=mode)
        got = O.preprocess_images(decoded, mode=mode)
        assert ref.shape == got.shape and torch.equal(ref, got), (mode, ref.shape, got.shape, (ref - got).abs().max())
        u8 = torch.round(ref * 255).to(torch.uint8)
        assert torch.equal(u8.float() / 255, ref)
        arrs[f"{mode}.out_u8"] = u8.numpy()
        for i, a in enumerate(decoded):
            arrs[f"{mode}.in{i}"] = a
        print(mode, [a.shape for a in decoded], "->", tuple(ref.shape), "oracle bit-exact")
np.savez_compressed(os.path.join(OUT, "g20_load_fn.npz"), **arrs)
print(os.path.getsize(os.path.join(OUT, "g20_load_fn.npz")) / 1e6, "MB")

"""Import harness for the READ-ONLY reference at /root/reference (build container only).

Used ONLY by tools/make_golden.py to (a) pin the oracle/ restatement against the
reference's own code and (b) emit the small fixtures committed under tests/golden/.
Nothing in tests/, bench.py or the package imports this file; /root/reference does
not exist on the GPU box.

Recipe follows SURVEY.md Appendix A: permissive stub modules for the reference's
missing third-party imports, with real semantics only where the hot path needs them
(torchvision tensor resize = bilinear, align_corners=False, no antialias).
"""
import importlib
import importlib.machinery
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"


class _Anything:
    """Placeholder object: attribute access / calls / subclassing all succeed."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    m.__path__ = []

    def _getattr(attr, _name=name):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Anything()

    m.__getattr__ = _getattr
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _tv_resize(img, size, *a, **k):
    # torchvision 0.16.2 tensor path (requirements.txt:16): bilinear, no antialias
    squeeze = False
    if img.dim() == 3:
        img, squeeze = img.unsqueeze(0), True
    out = F.interpolate(img, size=tuple(size), mode="bilinear", align_corners=False, antialias=False)
    return out.squeeze(0) if squeeze else out


class _LightningModule(nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, *a, **k):
        pass

    @property
    def device(self):
        return torch.device("cpu")


_installed = False


def install():
    global _installed
    if _installed:
        return
    _installed = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
        sys.path.insert(0, REF + "/dust3r")
        sys.path.insert(0, REF + "/dust3r/croco")
    for name in [
        "cv2", "visdom", "kornia", "kornia.filters", "kornia.morphology", "albumentations",
        "imageio", "pycocotools", "pycocotools.coco", "timm", "timm.data", "timm.models",
        "timm.models.layers", "timm.layers",
        "pytorch_lightning.loggers", "pytorch_lightning.callbacks", "matplotlib", "matplotlib.cm",
        "matplotlib.pyplot", "tensorboard", "roma", "xformers", "xformers.ops", "trimesh",
        "hydra.core", "hydra.core.hydra_config", "omegaconf",
    ]:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                _stub(name)
    tvf = _stub("torchvision.transforms.functional", resize=_tv_resize)

    class _ToTensor:
        """torchvision.transforms.ToTensor on a PIL RGB image: uint8 HWC -> float32 CHW / 255."""

        def __call__(self, pic):
            import numpy as np
            a = np.asarray(pic, dtype=np.uint8)
            if a.ndim == 2:
                a = a[:, :, None]
            return torch.from_numpy(a.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    tvt = _stub("torchvision.transforms", functional=tvf, ToTensor=_ToTensor)
    _stub("torchvision", transforms=tvt)
    pl = _stub("pytorch_lightning", LightningModule=_LightningModule, Callback=object)
    pl.loggers = sys.modules["pytorch_lightning.loggers"]
    pl.callbacks = sys.modules["pytorch_lightning.callbacks"]

    def _hydra_main(*a, **k):
        return lambda fn: fn

    _stub("hydra", main=_hydra_main)


def ref_utils():
    """(utils.losses, utils.functions, utils.model) of the reference."""
    install()
    import utils.losses as L
    import utils.functions as Fn
    import utils.model as M
    return L, Fn, M


def ref_modules():
    """(FinetuneMASt3RTIMM, FinetuneVGGTTIMM, FinetuneTIMM) classes of the reference."""
    install()
    mods = []
    for n in ("src.finetune_timm_mast3r", "src.finetune_timm_vggt", "src.finetune_timm_me"):
        try:
            mods.append(importlib.import_module(n))
        except Exception as e:  # pragma: no cover - reported by make_golden
            print(f"[ref_import] could not import {n}: {type(e).__name__}: {e}")
            mods.append(None)
    return mods


def ref_vit():
    install()
    from vggt.layers import vision_transformer as vt
    return vt

"""The reference's Hydra configs (config/finetune_timm_*.yaml) and the trainer registry / Trainer arguments of
src/main.py:78-82,127-158, mapped onto `FinetuneGD` — a user of the reference keeps the same yaml files.

A config is the five top-level keys the reference reads — `model`, `backbone`, `dataset`, `matcher`,
`evaluation_methods` (the `hydra:` block only names output directories and is ignored) — and selects:
  * the trainer: `model_{matcher}` -> FinetuneMASt3RTIMM | FinetuneVGGTTIMM | FinetuneTIMM (src/main.py:78-82,127)
    = FinetuneGD(variant = "mast3r" | "vggt" | "me") with that trainer's default loss weights / temperature schedule
    (src/finetune_timm_mast3r.py:73-86, src/finetune_timm_vggt.py:82-94);
  * the backbone: the only registered name is 'ViT-B-16' -> timm 'vit_base_patch16_clip_384.laion2b_ft_in12k_in1k'
    (src/finetune_timm_mast3r.py:68-70): a pre-norm CLIP ViT-B/16 (LN eps 1e-5, CLIP mean / std, no patch-conv bias);
    BASELINE.json's DINOv2-style /14 backbones are registered here beside it;
  * Trainer constants: r = 4, AdamW(lr 1e-5, wd 1e-4), gradient_clip_val 1.0, max_epochs 500 (src/main.py:127,147-158).
"""
import os

CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)

# backbone registry: name in the yaml -> (FinetuneGD backbone preset, patch, native image size, GDViT kwargs)
BACKBONES = {
    "ViT-B-16": ("vit_base", 16, 384, dict(pre_norm=True, ln_eps=1e-5, pos_interp="timm", mean=CLIP_MEAN, std=CLIP_STD)),
    "ViT-S-14": ("vit_small", 14, 518, dict(init_values=1.0)),
    "ViT-B-14": ("vit_base", 14, 518, dict(init_values=1.0)),
    "ViT-L-14": ("vit_large", 14, 518, dict(init_values=1.0)),
    "CLIP-ViT-L-14": ("vit_large", 14, 336, dict(pre_norm=True, ln_eps=1e-5, pos_interp="timm", mean=CLIP_MEAN, std=CLIP_STD)),
}

# src/main.py:78-82 — `model[f"{cfg.model}_{cfg.matcher}"]`
TRAINERS = {
    "timm_mast3r": dict(variant="mast3r", trainer="FinetuneMASt3RTIMM", ap_loss_weight=1.0, depth_loss_weight=0.0,
                        intra_depth_loss_weight=1.0, kl_loss_weight=1.0, init_temperature=1.0, final_temperature=0.5),
    "timm_vggt": dict(variant="vggt", trainer="FinetuneVGGTTIMM", ap_loss_weight=1.0, depth_loss_weight=1.0,
                      intra_depth_loss_weight=1.0, kl_loss_weight=1.0, init_temperature=1.0, final_temperature=1.0),
    "timm_me": dict(variant="me", trainer="FinetuneTIMM", ap_loss_weight=1.0, depth_loss_weight=0.0,
                    intra_depth_loss_weight=0.0, kl_loss_weight=0.0, init_temperature=1.0, final_temperature=1.0),
}
TRAINER_ARGS = dict(r=4, lr=1e-5, weight_decay=1e-4, gradient_clip_val=1.0, max_epochs=500)

# the five files the reference ships, as key / value dicts (config/finetune_timm_*.yaml:1-17; `hydra:` block dropped)
_EVAL3 = ["semantic_transfer", "tracking", "pose"]
PRESETS = {
    "finetune_timm_mast3r_objaverse": dict(model="timm", backbone="ViT-B-16", dataset="objaverse", matcher="mast3r", evaluation_methods=_EVAL3),
    "finetune_timm_mast3r_scannetpp": dict(model="timm", backbone="ViT-B-16", dataset="scannetpp", matcher="mast3r", evaluation_methods=_EVAL3),
    "finetune_timm_me_objaverse": dict(model="timm", backbone="ViT-B-16", dataset="objaverse", matcher="me", evaluation_methods=_EVAL3),
    "finetune_timm_vggt_objaverse": dict(model="timm", backbone="ViT-B-16", dataset="objaverse", matcher="vggt", evaluation_methods=_EVAL3),
    "finetune_timm_vggt_scannetpp": dict(model="timm", backbone="ViT-B-16", dataset="scannetpp", matcher="vggt",
                                         evaluation_methods=["semantic_transfer", "tracking"]),
}
KEYS = ("model", "backbone", "dataset", "matcher", "evaluation_methods")
DATASETS = {"objaverse": ("mast3r", "vggt", "me"), "scannetpp": ("mast3r", "vggt")}     # src/main.py:60-74 get_dataset


class ConfigError(ValueError):
    pass


def resolve(raw, name="config"):
    """The reference's keys -> a flat dict: the keys themselves + trainer selection + loss weights + Trainer constants."""
    missing = [k for k in KEYS[:4] if k not in raw]
    if missing:
        raise ConfigError(f"{name}: missing key(s) {missing} (the reference reads {KEYS})")
    key = f"{raw['model']}_{raw['matcher']}"
    if key not in TRAINERS:
        raise ConfigError(f"{name}: no trainer registered for model_matcher = '{key}' (src/main.py:78-82 has {sorted(TRAINERS)})")
    if raw["backbone"] not in BACKBONES:
        raise ConfigError(f"{name}: unknown backbone '{raw['backbone']}' (registered: {sorted(BACKBONES)})")
    if raw["dataset"] not in DATASETS or raw["matcher"] not in DATASETS[raw["dataset"]]:
        raise ConfigError(f"{name}: dataset '{raw['dataset']}' has no '{raw['matcher']}' loader (src/main.py:60-74)")
    cfg = {"name": name}
    cfg.update({k: raw[k] for k in KEYS if k in raw})
    cfg.setdefault("evaluation_methods", ["semantic_transfer"])        # src/main.py:95-97 default
    cfg.update(TRAINERS[key])
    cfg.update(TRAINER_ARGS)
    for k in ("ap_loss_weight", "depth_loss_weight", "intra_depth_loss_weight", "kl_loss_weight", "init_temperature",
              "final_temperature", "r", "lr", "weight_decay", "gradient_clip_val", "max_epochs", "batch_size"):
        if k in raw:            # Hydra command-line style overrides (`+kl_loss_weight=0.5`) land in the same dict
            cfg[k] = raw[k]
    return cfg


def preset(name):
    return resolve(PRESETS[name], name)


def load(path):
    """Read one of the reference's yaml files (or a file with the same keys)."""
    import yaml
    with open(path) as fh:
        raw = yaml.safe_load(fh) or {}
    raw.pop("hydra", None)
    return resolve(raw, os.path.splitext(os.path.basename(path))[0])


def engine_kwargs(cfg, backbone=None, img_size=None, **overrides):
    """FinetuneGD constructor arguments for a resolved config.  `backbone` overrides the yaml's registry name (e.g. the
    BASELINE's 'ViT-B-14' in place of the reference's only registered 'ViT-B-16')."""
    preset_name, patch, native, vk = BACKBONES[backbone if backbone in BACKBONES else cfg["backbone"]]
    if backbone is not None and backbone not in BACKBONES:       # a vit.VIT_PRESETS name (e.g. the tests' tiny ViT)
        preset_name = backbone
    kw = dict(r=cfg["r"], backbone=preset_name, patch_size=patch, img_size=img_size or native, variant=cfg["variant"],
              ap_loss_weight=cfg["ap_loss_weight"], depth_loss_weight=cfg["depth_loss_weight"],
              intra_depth_loss_weight=cfg["intra_depth_loss_weight"], kl_loss_weight=cfg["kl_loss_weight"],
              init_temperature=cfg["init_temperature"], final_temperature=cfg["final_temperature"],
              max_epochs=cfg["max_epochs"], vit_kwargs=dict(vk))
    kw.update(overrides)
    return kw


def build_engine(cfg, **overrides):
    """resolved config -> (FinetuneGD, optimizer arguments for FinetuneGD.configure_optimizers)."""
    from .finetune import FinetuneGD
    eng = FinetuneGD(**engine_kwargs(cfg, **overrides))
    return eng, dict(lr=cfg["lr"], weight_decay=cfg["weight_decay"], max_norm=cfg["gradient_clip_val"])

"""CPU: the reference's Hydra keys (config/finetune_timm_*.yaml, committed as key/value dicts in gd_amd.config.PRESETS)
resolve to the trainer / loss weights / Trainer constants of src/main.py, a yaml file with those keys loads, bad keys raise,
and the teacher temperature schedule follows src/finetune_timm_mast3r.py:217-224."""
import pytest
import torch


def test_presets_resolve_like_the_reference_registry():
    import gd_amd  # noqa: F401
    from gd_amd import config
    assert sorted(config.PRESETS) == ["finetune_timm_mast3r_objaverse", "finetune_timm_mast3r_scannetpp",
                                      "finetune_timm_me_objaverse", "finetune_timm_vggt_objaverse", "finetune_timm_vggt_scannetpp"]
    c = config.preset("finetune_timm_mast3r_scannetpp")
    assert (c["trainer"], c["variant"], c["depth_loss_weight"], c["final_temperature"]) == ("FinetuneMASt3RTIMM", "mast3r", 0.0, 0.5)
    assert (c["r"], c["lr"], c["weight_decay"], c["gradient_clip_val"], c["max_epochs"]) == (4, 1e-5, 1e-4, 1.0, 500)
    c = config.preset("finetune_timm_vggt_scannetpp")
    assert (c["trainer"], c["depth_loss_weight"], c["final_temperature"]) == ("FinetuneVGGTTIMM", 1.0, 1.0)
    assert c["evaluation_methods"] == ["semantic_transfer", "tracking"]
    assert config.preset("finetune_timm_me_objaverse")["trainer"] == "FinetuneTIMM"
    kw = config.engine_kwargs(config.preset("finetune_timm_vggt_objaverse"))
    assert kw["patch_size"] == 16 and kw["vit_kwargs"]["pre_norm"] and kw["vit_kwargs"]["ln_eps"] == 1e-5      # CLIP ViT-B/16
    kw = config.engine_kwargs(config.preset("finetune_timm_vggt_objaverse"), backbone="ViT-B-14", img_size=518)
    assert (kw["backbone"], kw["patch_size"], kw["img_size"]) == ("vit_base", 14, 518)


def test_yaml_file_with_the_reference_keys_loads(tmp_path):
    import gd_amd  # noqa: F401
    from gd_amd import config
    f = tmp_path / "finetune_timm_vggt_scannetpp.yaml"
    f.write_text("model: timm\nbackbone: ViT-B-16\ndataset: scannetpp\nmatcher: vggt\n\nhydra:\n  run:\n    dir: outputs/x\n\n"
                 "evaluation_methods:\n  - semantic_transfer\n  - tracking\n")
    assert config.load(str(f)) == config.preset("finetune_timm_vggt_scannetpp")
    g = tmp_path / "custom.yaml"
    g.write_text("model: timm\nbackbone: ViT-L-14\ndataset: objaverse\nmatcher: mast3r\nkl_loss_weight: 0.25\n")
    c = config.load(str(g))
    assert c["kl_loss_weight"] == 0.25 and c["evaluation_methods"] == ["semantic_transfer"] and c["name"] == "custom"
    for bad in ("model: timm\nbackbone: ViT-B-16\ndataset: scannetpp\nmatcher: me\n",          # no ME loader for scannetpp
                "model: timm\nbackbone: ViT-Z\ndataset: objaverse\nmatcher: vggt\n",
                "model: dino\nbackbone: ViT-B-16\ndataset: objaverse\nmatcher: vggt\n",
                "backbone: ViT-B-16\n"):
        h = tmp_path / "bad.yaml"
        h.write_text(bad)
        with pytest.raises(config.ConfigError):
            config.load(str(h))


def test_engine_from_config_and_temperature_schedule():
    import gd_amd  # noqa: F401
    from gd_amd import config
    cfg = config.preset("finetune_timm_mast3r_objaverse")
    eng, opt = config.build_engine(cfg, backbone="vit_tiny_test", patch_size=14, img_size=56, dtype="f32", vit_kwargs={})
    assert opt == dict(lr=1e-5, weight_decay=1e-4, max_norm=1.0)
    assert (eng.variant, eng.depth_loss_weight, eng.kl_loss_weight) == ("mast3r", 0.0, 1.0)
    assert eng.teacher_temperature == 1.0
    assert eng.update_temperature(current_epoch=250) == pytest.approx(0.75)        # 1.0 -> 0.5 linearly over 500 epochs
    assert eng.update_temperature(current_epoch=10_000) == pytest.approx(0.5)
    eng.current_epoch = 125
    eng.on_train_batch_end()
    assert eng.teacher_temperature == pytest.approx(0.875)
    engv, _ = config.build_engine(config.preset("finetune_timm_vggt_objaverse"), backbone="vit_tiny_test", patch_size=14,
                                  img_size=56, dtype="f32", vit_kwargs={})
    assert engv.update_temperature(current_epoch=400) == 1.0 and engv.depth_loss_weight == 1.0
    # the optimiser skips the never-used depth_attention span
    flat = eng.configure_optimizers(**opt)
    n_skip = sum((q.numel() + 3) // 4 * 4 for q in eng.depth_diff_head.depth_attention.parameters())
    assert sum(b - a for a, b in flat["live"]) == flat["p"].numel() - n_skip and len(flat["live"]) == 2

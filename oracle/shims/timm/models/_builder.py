def build_model_with_cfg(model_cls, variant, pretrained=False, **kwargs):
    kwargs.pop("pretrained_cfg", None)
    kwargs.pop("pretrained_cfg_overlay", None)
    kwargs.pop("features_only", None)
    kwargs.pop("pretrained_strict", None)
    return model_cls(**kwargs)

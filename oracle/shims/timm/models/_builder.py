"""Shim of timm.models._builder.build_model_with_cfg: with pretrained=False upstream reduces to
constructing the class from the model kwargs; the pretrained-config / filter arguments only matter
for checkpoint download (tiny_vit.py:621-653)."""


def build_model_with_cfg(model_cls, variant, pretrained=False, pretrained_cfg=None, default_cfg=None,
                         pretrained_filter_fn=None, **kwargs):
    if pretrained:
        raise RuntimeError("shim: pretrained weights cannot be downloaded (no network)")
    return model_cls(**kwargs)

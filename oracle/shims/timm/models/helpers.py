from ._builder import build_model_with_cfg  # noqa: F401

"""Shim: placeholder for efficientvit/dc_ae.py's import (off the hot path)."""
MISSING = "???"


class OmegaConf:
    @staticmethod
    def create(*a, **k):
        return {}

    @staticmethod
    def structured(*a, **k):
        return {}

    @staticmethod
    def merge(*a, **k):
        return {}

    @staticmethod
    def to_object(x):
        return x

    @staticmethod
    def from_dotlist(x):
        return {}


class DictConfig(dict):
    pass

"""Shim placeholder (off the hot path)."""

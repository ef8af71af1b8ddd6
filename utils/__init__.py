"""Alias package: the reference's `utils.types` / `utils.utils` import paths (utils/types.py:8-31)."""

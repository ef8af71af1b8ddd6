"""`utils.types` of the reference (utils/types.py:8-31): the dataclasses crossing the text-encoder seam."""
from view_neti_amd.compat.types import MapperOutput, NeTIBatch, PESigmas  # noqa: F401

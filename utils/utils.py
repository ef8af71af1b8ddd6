"""`utils.utils` of the reference (utils/utils.py:5-16)."""
from view_neti_amd.compat.utils_utils import num_to_string, string_to_num  # noqa: F401

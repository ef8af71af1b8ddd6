from view_neti_amd.compat.neti_modules import NeTIMapper  # noqa: F401

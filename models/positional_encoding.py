from view_neti_amd.compat.neti_modules import FourierPositionalEncodingNDims  # noqa: F401

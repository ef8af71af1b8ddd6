from view_neti_amd.compat.neti_modules import FourierPositionalEncodingNDims  # noqa: F401
from view_neti_amd.compat.neti_modules import NeTIPositionalEncoding  # noqa: F401

"""alias of the reference module path (sd_pipeline_call.py at the repository root)."""
from view_neti_amd.compat.sd_pipeline_call import *  # noqa: F401,F403
from view_neti_amd.compat.sd_pipeline_call import sd_pipeline_call, get_neg_prompt_input_ids  # noqa: F401

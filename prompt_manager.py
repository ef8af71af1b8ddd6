"""alias of the reference module path (prompt_manager.py at the repository root)."""
from view_neti_amd.compat.prompt_manager import PromptManager, PromptEmbeds  # noqa: F401

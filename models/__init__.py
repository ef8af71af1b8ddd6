"""Alias package: reference checkpoints pickle `models.positional_encoding.*` / `models.neti_mapper.*`
class paths (checkpoint_handler.py:70-71,86); the implementations live in view_neti_amd.compat."""

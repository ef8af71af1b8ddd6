"""Reference-compatible surface modules (config, checkpoint handler, small helpers)."""

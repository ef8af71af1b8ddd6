"""Lookup table key -> path of released view mappers (the reference's training/pretrained_models.py
data table, kept for config-surface parity: `model.pretrained_view_mapper_key`)."""
lookup_pretrained_models = {
    "0": None,
    "1": "results/mode3_4scenes/mapper-steps-50000_view.pt",
    "8": "results/mapper-steps-50000_view.pt",
}

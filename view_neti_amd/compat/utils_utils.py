"""Small string/number helpers with the semantics of the reference's utils/utils.py:5-24
(view-token strings use 'p' for the decimal point)."""
from pathlib import Path


def num_to_string(num, tol: int = 2) -> str:
    # integers print without decimals; anything else is rounded to `tol` places, '.' -> 'p'
    if int(num) == num:
        return str(int(num))
    return f"{num:.{tol}f}".replace(".", "p")


def string_to_num(s: str) -> float:
    return float(s.replace("p", "."))


def parameters_checksum(model) -> float:
    """sum of |p| over all parameters (reference utils/utils.py:27-33)."""
    if model is None:
        return 0
    return sum(p.abs().sum().item() for p in model.parameters())


def filter_paths_imgs(paths):
    return [p for p in paths if Path(p).suffix in (".png", ".jpg")]

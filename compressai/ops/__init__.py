from .bound_ops import LowerBound
from .ops import ste_round
from .parametrizers import NonNegativeParametrizer

__all__ = ["ste_round", "LowerBound", "NonNegativeParametrizer"]

from .entropy_models import (EntropyBottleneck, EntropyModel, GaussianConditional, GaussianMixtureConditional,
                             GaussianMixtureConditional_gf)

__all__ = ["EntropyModel", "EntropyBottleneck", "GaussianConditional", "GaussianMixtureConditional",
           "GaussianMixtureConditional_gf"]

"""`compressai` API surface for the MI355X-native MASIC path.

Only the names the MASIC codec and its unchanged drivers import are provided (reference:
compressai/__init__.py:30-77, and the import lists of coremasic/mywork/MASIC.py:18-38 and
newtrain_codec_real.py:8-30).  The compute behind them is the HIP library in `masic_amd`; there is
no ATen/CPU fallback.  Sub-packages of the upstream model zoo that MASIC never touches
(zoo, transforms, utils, models.google/waseda/video) are out of scope.
"""
_entropy_coder = "ans"
_available_entropy_coders = [_entropy_coder]


def set_entropy_coder(entropy_coder):
    """reference compressai/__init__.py:52-63"""
    global _entropy_coder
    if entropy_coder not in _available_entropy_coders:
        raise ValueError(f'Invalid entropy coder "{entropy_coder}", choose from ({", ".join(_available_entropy_coders)}).')
    _entropy_coder = entropy_coder


def get_entropy_coder():
    """reference compressai/__init__.py:66-70"""
    return _entropy_coder


def available_entropy_coders():
    """reference compressai/__init__.py:73-77"""
    return _available_entropy_coders


from compressai import ops, layers, models, entropy_models, datasets, ans, _CXX  # noqa: E402,F401

__version__ = "1.2.0b3.masic_amd"

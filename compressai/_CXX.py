"""`compressai._CXX` (reference: pybind11 module from compressai/cpp_exts/ops/ops.cpp:108-118)."""
from masic_amd import rans as _rans


def pmf_to_quantized_cdf(pmf, precision):
    """list of float -> list of int (reference ops.cpp:41-106)"""
    return _rans.pmf_to_quantized_cdf(pmf, precision).tolist()

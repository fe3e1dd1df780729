from .utils import conv, deconv, update_registered_buffers  # noqa: F401
from .base import CompressionModel  # noqa: F401

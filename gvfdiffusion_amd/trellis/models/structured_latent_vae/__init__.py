from .base import SparseTransformerBase  # noqa: F401
from .decoder_gs import SLatGaussianDecoder  # noqa: F401

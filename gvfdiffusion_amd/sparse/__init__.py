"""Sparse-voxel side of the path (secondary rows SP1-SP5): the SparseTensor container,
vox2seq voxel serialisation, the full / windowed / serialized sparse attention operators and the row-wise layers
(SparseLinear, activations, norms, down / up-sampling) of the reference's `sparse/` package."""
from . import vox2seq  # noqa: F401
from .basic import *   # noqa: F401,F403
from .linear import SparseLinear  # noqa: F401
from .nonlinearity import *  # noqa: F401,F403
from .norm import *  # noqa: F401,F403
from .spatial import *  # noqa: F401,F403


def __getattr__(name):
    # attention lives in a sub-package that imports the kernels' op wrappers: resolve lazily, as the reference does
    # (sparse/__init__.py:68-80)
    if name in ("sparse_scaled_dot_product_attention", "SerializeMode", "SerializeModes",
                "sparse_serialized_scaled_dot_product_self_attention", "sparse_windowed_scaled_dot_product_self_attention",
                "SparseMultiHeadAttention", "SparseMultiHeadRMSNorm"):
        from . import attention
        return getattr(attention, name)
    raise AttributeError(f"module {__name__} has no attribute {name}")

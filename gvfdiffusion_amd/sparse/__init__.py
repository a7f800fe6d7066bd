"""Sparse-voxel side of the path (secondary rows SP1-SP5): the SparseTensor container,
vox2seq voxel serialisation, and the full / windowed / serialized sparse attention operators."""
from . import vox2seq  # noqa: F401
from .basic import *   # noqa: F401,F403

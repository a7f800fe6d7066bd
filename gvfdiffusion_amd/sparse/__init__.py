"""Sparse-voxel side of the path (secondary rows SP1-SP5)."""
from . import vox2seq  # noqa: F401

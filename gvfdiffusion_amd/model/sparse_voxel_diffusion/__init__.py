"""Static-VAE backbone over sparse voxels (model/sparse_voxel_diffusion/ of the reference; SURVEY.md section 8f NEXT #3)."""
from .sparse_transformer import AbsolutePositionEmbedder, SparseFeedForward, SparseTransformerBlock, block_attn_config  # noqa: F401
from .sparse_transformer_vae import SparseTransformerVAE  # noqa: F401
from .sparse_vae import SparseVAE  # noqa: F401

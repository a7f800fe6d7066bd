from .structured_latent_vae import SLatGaussianDecoder, SparseTransformerBase  # noqa: F401

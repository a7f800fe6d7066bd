"""The one TRELLIS component next to the hot path: the structured-latent Gaussian decoder
(trellis/models/structured_latent_vae/decoder_gs.py), whose output is the canonical Gaussian set every frame of the path
deforms.  The image -> structured-latent generator (spconv flow models, DINOv2 conditioning) is out of scope."""

"""The repo's inference_dpm_latent.py (the hot-path half of the reference's entry script, inference_dpm_latent.py:41-273) end to end at a
small size: synthetic weights of the released architectures, FPS conditions, DPM-Solver over the DiT, de-normalise, VAE decode, batched
render, the one frame gather -- the same flags the reference's script takes."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BASE = ["--synthetic", "--num_samples", "2", "--num_timesteps", "4", "--rescale_timesteps", "4", "--gaussians", "4096", "--resolution", "64",
        "--views", "2"]


def test_flags_of_the_reference_script_are_accepted():
    import inference_dpm_latent as S
    a = S.create_argparser().parse_args(["--exp_name", "/tmp/x", "--ckpt", "a.pt", "--vae_ckpt", "b.pt", "--batch_size", "1", "--seed", "3", "--use_fp16",
                                         "--config", "configs/diffusion.yml", "--deformation_mean_file", "None", "--deformation_std_file", "m.pt",
                                         "--num_timesteps", "32", "--num_samples", "4", "--rescale_timesteps", "100", "--guidance_scale", "2.0",
                                         "--guidance_scale2", "1.5", "--adaptive"])
    assert a.deformation_mean_file is None and a.deformation_std_file == "m.pt" and a.adaptive and a.use_fp16 and a.num_timesteps == 32
    model_cfg, diff_cfg, vae_cfg = S._load_config("does/not/exist.yml")
    assert model_cfg["model_channels"] == 512 and vae_cfg["num_latents"] == 512 and diff_cfg["predict_type"] == "v"


@pytest.mark.gpu
def test_script_runs_the_chain_and_is_reproducible(cuda, tmp_path):
    import inference_dpm_latent as S
    f1 = S.main(BASE)
    assert f1.shape == (2, 4 * 2, 3, 64, 64) and f1.dtype == torch.uint8 and f1.is_cuda
    assert not torch.equal(f1[0], f1[1]) and int(f1.float().std()) > 0               # two different samples, not blank frames
    assert (f1[0, 0].float() - f1[0, 2].float()).abs().max() > 0                     # the decoded deltas move the object between timesteps
    # the multistep solver announces its grid through the wrapper the script hands it: every evaluation after the announcement takes its
    # modulation from the precomputed table (the script's counting closure used to hide `prepare_times`, bench.py measured a path
    # the product script did not run)
    dit = S.main.last_chain.models[0][0]
    assert dit.__dict__.get("mod_table_hits", 0) >= 2 * 4, dit.__dict__.get("mod_table_hits", 0)
    f2 = S.main(BASE + ["--in_flight", "2"])                                        # two samples in flight, own model copies: same frames
    assert torch.equal(f1, f2)
    f3 = S.main(BASE + ["--use_fp16", "--adaptive", "--rescale_timesteps", "100", "--save_png", "--exp_name", str(tmp_path)])
    assert f3.shape == f1.shape
    pngs = sorted(os.listdir(tmp_path / "inference_images"))
    assert len(pngs) == 2 * 4 * 2 and pngs[0] == "rank_00_render_000000_cam_000_timesteps_00.png"      # the reference's file names (:297)
    with pytest.raises(SystemExit):
        S.main(["--num_samples", "1"])                                               # neither checkpoints nor --synthetic


@pytest.mark.gpu
def test_two_samples_in_flight_with_a_capture_per_sample_equal_serial_sampling(cuda, monkeypatch):
    """Full-size chains (configs/diffusion.yml DiT, adaptive DPM-Solver, motion-VAE decode, render), two in flight, every sample a NEW hipGraph
    capture on its slot's thread while the other slot samples, decodes and renders: latents, decoded deltas and uint8 frames bit-identical
    to the same samples computed one after the other.  Rounds 3-4 failed this in 25-50 % of the rounds (packed-fp32 arithmetic of one
    kernel beside another kernel's MFMAs on the same CU: profiles/r04_inflight_root_cause.txt); the library is built without packed fp32 now."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    monkeypatch.setenv("REPRO_ROUNDS", "10")
    monkeypatch.setenv("REPRO_SAMPLES", "4")
    for k in ("REPRO_EAGER", "REPRO_PAUSE", "REPRO_DET", "REPRO_NO_EMPTY", "REPRO_NO_GC", "REPRO_NO_SYNC", "REPRO_SOLOCOND"):
        monkeypatch.delenv(k, raising=False)
    import inflight_capture_repro as H
    out = H.main()
    assert out["rounds"] == 10 and out["divergent_rounds"] == 0 and out["divergent_samples"] == 0, out

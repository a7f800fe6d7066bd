"""Frame post-process (utils/inference_utils.py:276-297 of the reference: PIL LANCZOS resize -> pad / crop to 512):
the numpy oracle and the product's coefficient tables against Pillow itself (CPU), the HIP passes against Pillow (GPU).
All comparisons are bit-exact (uint8)."""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle import resize_ref


def pil_resize(img_hw3, target):
    return np.asarray(Image.fromarray(img_hw3).resize((target, target), resample=Image.Resampling.LANCZOS))


def pil_pad_crop(img_hw3, target, out=512):
    """The reference's lines 283-296, verbatim in behaviour."""
    image = Image.fromarray(img_hw3).resize((target, target), resample=Image.Resampling.LANCZOS)
    W, H = image.size
    if H < out or W < out:
        new = Image.new("RGB", (out, out), (255, 255, 255))
        new.paste(image, (max(0, (out - W) // 2), max(0, (out - H) // 2)))
        return np.asarray(new)
    left, top = (W - out) // 2, (H - out) // 2
    return np.asarray(image.crop((left, top, left + out, top + out)))


def frames(n, S, seed):
    g = np.random.default_rng(seed)
    f = g.integers(0, 256, size=(n, 3, S, S), dtype=np.uint8)
    f[0, :, S // 4:S // 2] = 255          # saturated and black bands: the filter's negative lobes clip at both ends
    f[0, :, S // 2:3 * S // 4] = 0
    return f


@pytest.mark.parametrize("S,target", [(100, 64), (100, 100), (100, 137), (64, 20), (50, 333), (97, 96)])
def test_oracle_matches_pillow(S, target):
    f = frames(2, S, S + target)
    got = resize_ref.resize_lanczos(f, target, target)
    for i in range(2):
        want = pil_resize(np.ascontiguousarray(f[i].transpose(1, 2, 0)), target).transpose(2, 0, 1)
        assert np.array_equal(got[i], want)


def test_product_tables_equal_the_oracle_tables():
    from gvfdiffusion_amd.utils.image_ops import resample_table
    for n_in, n_out in ((800, 512), (800, 614), (800, 409), (100, 137), (64, 20)):
        first, count, coef, ksize = resample_table(n_in, n_out)
        tab = resize_ref.coeffs(n_in, n_out)
        assert ksize == int(np.ceil(3.0 * max(n_in / n_out, 1.0))) * 2 + 1
        for x, (xmin, k) in enumerate(tab):
            assert first[x] == xmin and count[x] == len(k)
            assert [coef[j][x] for j in range(len(k))] == k.tolist()
            assert all(coef[j][x] == 0 for j in range(len(k), ksize))


@pytest.mark.parametrize("target", [300, 512, 700])
def test_oracle_pad_crop_matches_the_reference_lines(target):
    f = frames(1, 200, target)
    want = pil_pad_crop(np.ascontiguousarray(f[0].transpose(1, 2, 0)), target).transpose(2, 0, 1)
    assert np.array_equal(resize_ref.resize_pad_crop(f, target)[0], want)


@pytest.mark.gpu
@pytest.mark.parametrize("S,target,out", [(800, 512, 512), (800, 409, 512), (800, 614, 512), (800, 800, 512), (160, 160, 160),
                                          (123, 77, 64), (96, 301, 128)])
def test_device_matches_pillow(cuda, S, target, out):
    from gvfdiffusion_amd.utils.image_ops import resize_pad_crop_u8
    f = frames(3, S, S * 7 + target)
    got = resize_pad_crop_u8(torch.from_numpy(f).cuda(), target, out_size=out).cpu().numpy()
    assert got.shape == (3, 3, out, out)
    for i in range(3):
        want = pil_pad_crop(np.ascontiguousarray(f[i].transpose(1, 2, 0)), target, out).transpose(2, 0, 1)
        assert np.array_equal(got[i], want), (i, int(np.abs(got[i].astype(int) - want.astype(int)).max()))


@pytest.mark.gpu
def test_device_other_filters_and_loud_failures(cuda):
    from gvfdiffusion_amd._lib import GvfError
    from gvfdiffusion_amd.utils.image_ops import resize_pad_crop_u8
    f = frames(2, 120, 5)
    for name, pil in (("bicubic", Image.Resampling.BICUBIC), ("bilinear", Image.Resampling.BILINEAR), ("box", Image.Resampling.BOX)):
        got = resize_pad_crop_u8(torch.from_numpy(f).cuda(), 75, out_size=75, filt=name).cpu().numpy()
        want = np.asarray(Image.fromarray(np.ascontiguousarray(f[1].transpose(1, 2, 0))).resize((75, 75), resample=pil)).transpose(2, 0, 1)
        assert np.array_equal(got[1], want), name
    with pytest.raises(GvfError):
        resize_pad_crop_u8(torch.from_numpy(f), 75)                     # host tensor: no CPU fallback
    with pytest.raises(ValueError):
        resize_pad_crop_u8(torch.from_numpy(f).cuda().float(), 75)


@pytest.mark.gpu
def test_device_empty_batch_and_single_row_images(cuda):
    from gvfdiffusion_amd.utils.image_ops import resize_pad_crop_u8
    out = resize_pad_crop_u8(torch.zeros((0, 3, 40, 40), dtype=torch.uint8, device=cuda), 20, out_size=32)
    assert out.shape == (0, 3, 32, 32)
    f = frames(1, 9, 1)                                        # smaller than the filter support in both axes
    got = resize_pad_crop_u8(torch.from_numpy(f).cuda(), 3, out_size=8).cpu().numpy()
    want = pil_pad_crop(np.ascontiguousarray(f[0].transpose(1, 2, 0)), 3, 8).transpose(2, 0, 1)
    assert np.array_equal(got[0], want)

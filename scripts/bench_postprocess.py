"""Times the device-side frame post-process (uint8 LANCZOS resize + pad / crop to 512x512, csrc/resize.hip) on a chunk of
800x800 frames, next to Pillow doing the same per frame on the host (the reference's utils/inference_utils.py:276-296)."""
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvfdiffusion_amd.utils.image_ops import resize_pad_crop_u8  # noqa: E402


def main():
    F, S = 96, 800
    frames = torch.randint(0, 256, (F, 3, S, S), dtype=torch.uint8, device="cuda")
    for target in (409, 512, 614):
        out = resize_pad_crop_u8(frames, target)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            resize_pad_crop_u8(frames, target, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        alg = F * 3 * (S * S + S * target + S * target + 512 * 512)        # read src, write + read the intermediate, write the canvas
        host = frames[:4].permute(0, 2, 3, 1).contiguous().cpu().numpy()
        t0 = time.perf_counter()
        for k in range(4):
            im = Image.fromarray(host[k]).resize((target, target), resample=Image.Resampling.LANCZOS)
            if target < 512:
                c = Image.new("RGB", (512, 512), (255, 255, 255)); c.paste(im, ((512 - target) // 2,) * 2)
            else:
                c = im.crop(((target - 512) // 2, (target - 512) // 2, (target - 512) // 2 + 512, (target - 512) // 2 + 512))
        pil_ms = (time.perf_counter() - t0) / 4 * 1e3
        print(f"target {target}: {ms / F * 1e3:.1f} us / frame on the device ({alg / ms / 1e6:.0f} GB/s of algorithmic bytes), "
              f"Pillow {pil_ms:.2f} ms / frame on one host core")


if __name__ == "__main__":
    main()

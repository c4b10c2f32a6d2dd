"""GPU: frame preprocessing kernels (row a3) against fixtures produced by Pillow / torch here (tools/make_goldens.py
preprocess). Resized uint8 images: bit-exact with PIL.Image.resize. Float stages: same operation order, fp32."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))


def _frame(H, W, seed):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    base = ((xx * 3 + yy * 5 + (xx * yy) % 7 * 9) % 256)[..., None] + np.array([0, 17, 34])
    return ((base + rng.randint(0, 64, size=(H, W, 3))) % 256).astype(np.uint8)


@pytest.mark.parametrize("interp", ["bicubic", "bilinear"])
def test_pil_resize_bit_exact(dev, interp):
    from merv_amd.preprocess import HALF_MEAN, HALF_STD, PILResizeNormalize
    z = np.load(G / "preprocess.npz")
    tr = PILResizeNormalize(interp, HALF_MEAN, HALF_STD)
    for i, (H, W) in enumerate(z["sizes"]):
        img = torch.from_numpy(_frame(int(H), int(W), 100 + i)).permute(2, 0, 1)[None].contiguous().to(dev)  # [1,3,H,W]
        _, u8 = tr(img, return_uint8=True)
        ref = torch.from_numpy(z[f"img{i}_{interp}_u8"])
        diff = (u8[0].cpu().int() - ref.int()).abs()
        assert int(diff.max()) == 0, (interp, int(H), int(W), int((diff > 0).sum()), int(diff.max()))


def test_pil_normalize_matches_torch_ops(dev):
    from merv_amd.preprocess import IMAGENET_MEAN, IMAGENET_STD, PILResizeNormalize
    z = np.load(G / "preprocess.npz")
    H, W = z["sizes"][1]
    frames = torch.from_numpy(np.stack([_frame(int(H), int(W), 101)] * 3, 0)).permute(0, 3, 1, 2).contiguous().to(dev)  # 3 equal frames
    out = PILResizeNormalize("bicubic", IMAGENET_MEAN, IMAGENET_STD)(frames)
    assert out.shape == (3, 3, 224, 224) and out.dtype == torch.float32
    ref = torch.from_numpy(z["img1_dinov2_pix_rows"])
    assert torch.equal(out[2, :, ::16].cpu(), ref)  # (u/255 - mean)/std with the same three fp32 roundings
    bf = PILResizeNormalize("bicubic", IMAGENET_MEAN, IMAGENET_STD, dtype=torch.bfloat16)(frames)
    assert torch.equal(bf.float(), out.to(torch.bfloat16).float())


def test_languagebind_transform(dev):
    from merv_amd.preprocess import LanguageBindTransform
    z = np.load(G / "preprocess.npz")
    for j, (H, W) in enumerate(z["lb_sizes"]):
        frames = np.stack([_frame(int(H), int(W), 200 + 10 * j + f) for f in range(2)], 0)  # [T,H,W,3]
        video = torch.from_numpy(frames).permute(0, 3, 1, 2).contiguous().to(dev)  # [T,3,H,W] as load_video returns
        out = LanguageBindTransform()(video)
        ref = torch.from_numpy(z[f"lb{j}_out"])
        assert out.shape == ref.shape == (3, 2, 224, 224)
        assert (out.cpu() - ref).abs().max() < 2e-5, float((out.cpu() - ref).abs().max())
        flipped = LanguageBindTransform(flip=True)(video)
        assert torch.equal(flipped.cpu(), out.cpu().flip(-1))


def test_transform_feeds_encoder_layouts(dev):
    from merv_amd.encoder import merv_full_specs
    from merv_amd.preprocess import transform_for
    video = torch.randint(0, 256, (32, 3, 120, 160), dtype=torch.uint8, device=dev)
    for spec in merv_full_specs():
        sub = video[:: 32 // spec.frames].contiguous()  # merv.py:803-806
        pix = transform_for(spec.name, torch.bfloat16)(sub)
        assert tuple(pix[None].shape) == spec.pixel_shape(1), spec.name
    with pytest.raises(ValueError):
        transform_for("dinov2")(video.float())

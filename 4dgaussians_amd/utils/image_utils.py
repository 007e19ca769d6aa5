"""utils/image_utils.py:14-42 of the reference (evaluation metrics of the FPS / eval harness)."""
import torch


@torch.no_grad()
def get_mse(img1, img2):
    return ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)


@torch.no_grad()
def psnr(img1, img2, mask=None):
    if mask is not None:
        img1, img2 = img1.flatten(1), img2.flatten(1)
        mask = torch.where(mask.flatten(1).repeat(3, 1) != 0, True, False)
        img1, img2 = img1[mask], img2[mask]
    mse = get_mse(img1, img2)
    out = 20 * torch.log10(1.0 / torch.sqrt(mse.float()))
    if mask is not None and torch.isinf(out).any():
        out = out[~torch.isinf(out)]
    return out

"""Device-side batch assembly: the tail of the reference's input pipeline (first slice).

The reference normalises every image on the host (ToTensor + Normalize, src/datasets/transforms_depth.py:618-660,
src/datasets/coco.py:76-79), converts depth / segmentation there (src/datasets/glassrgbd_norhint.py:277-281), pads and
masks in collate_fn_aux (src/util/misc.py:273-313) and ships 24 bytes per pixel of fp32 / int64 to the GPU.  Here the host
ships what the decoder produced - uint8 RGB, 16-bit depth in millimetres, uint8 labels: 5 bytes per pixel (8 with depth
widened to int32) - and ONE kernel (gwd_collate) writes the normalised pixel-major image batch, the padding mask, metric depth
and {0,1} labels.  Geometric / photometric augmentation (PIL resize, polygon clipping, colour jitter) stays where it is.
"""
import torch

from . import hip

MEAN, STD = (0.538, 0.494, 0.453), (0.257, 0.263, 0.273)          # src/datasets/coco.py:78


def device_collate(samples, device="cuda", dtype=torch.float32, mean=MEAN, std=STD):
    """samples: list (<= 16) of (rgb uint8 (h,w,3), depth_mm integer (h,w), labels uint8 (h,w)) host or device tensors, as
    decoded (any element but rgb may be None for the whole batch).  Returns the batch dict TrainStep / evaluate take:
    images (B,3,H,W) [a view of the pixel-major buffer the model reads in place], pad_mask (B,H,W) bool, depth (B,1,H,W) fp32
    metres, seg (B,1,H,W) int64."""
    lib = hip.library()
    if not 0 < len(samples) <= hip.COLLATE_BATCH:
        raise ValueError("1..%d images per call" % hip.COLLATE_BATCH)
    dev = torch.device(device)

    def up(t, dt):
        if t is None:
            return None
        t = torch.as_tensor(t)
        if t.device != dev and t.device.type == "cpu" and dev.type == "cuda":
            t = t.pin_memory()
        return t.to(dev, dtype=dt, non_blocking=True).contiguous()

    dev_samples = [(up(r, torch.uint8), up(d, torch.int32), up(l, torch.uint8)) for r, d, l in samples]
    for r, d, l in dev_samples:
        if r.dim() != 3 or r.shape[2] != 3 or (d is not None and d.shape != r.shape[:2]) or (l is not None and l.shape != r.shape[:2]):
            raise ValueError("rgb must be (h,w,3) with depth / labels of the same (h,w)")
    B = len(dev_samples)
    H, W = max(s[0].shape[0] for s in dev_samples), max(s[0].shape[1] for s in dev_samples)
    have_d, have_l = dev_samples[0][1] is not None, dev_samples[0][2] is not None
    images = torch.empty((B, H, W, 3), dtype=dtype, device=dev)
    mask = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    depth = torch.empty((B, H, W), dtype=torch.float32, device=dev) if have_d else None
    seg = torch.empty((B, H, W), dtype=torch.int64, device=dev) if have_l else None
    lib.collate(dev_samples, H, W, mean, std, images, mask, depth, seg)
    out = {"images": images.permute(0, 3, 1, 2), "pad_mask": mask.to(torch.bool)}
    if have_d:
        out["depth"] = depth.view(B, 1, H, W)
    if have_l:
        out["seg"] = seg.view(B, 1, H, W)
    return out

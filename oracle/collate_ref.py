"""CPU restatement (torch) of the tail of the reference's input pipeline.  TEST INFRASTRUCTURE ONLY.

* to_tensor / normalize: torchvision.transforms.functional (torchvision is NOT installed here and unpinned by the reference;
  call sites src/datasets/transforms_depth.py:618-660, constants src/datasets/coco.py:76-79) - restated from the published
  definitions: uint8 HWC -> float32 CHW `.div(255)`; `(t - mean[:,None,None]) / std[:,None,None]` in fp32.  *Parity unpinned
  by a reference-owned vector for these two lines alone* (they are plain IEEE fp32 ops);
* dataset tail: depth_gt / 1000.0, where(seg > 0, 1, 0).long() (src/datasets/glassrgbd_norhint.py:277-281);
* collate_fn_aux / nested_tensor_from_tensor_list (src/util/misc.py:273-313): restated AND pinned - tests/golden/collate.npz
  is produced by the reference's own collate_fn_aux (oracle/make_golden_collate.py).
"""
import torch

MEAN, STD = (0.538, 0.494, 0.453), (0.257, 0.263, 0.273)          # src/datasets/coco.py:78


def to_tensor(rgb_u8):
    """(h,w,3) uint8 -> (3,h,w) float32 in [0,1]."""
    return rgb_u8.permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def normalize(t, mean=MEAN, std=STD):
    m = torch.as_tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.as_tensor(std, dtype=torch.float32)[:, None, None]
    return (t - m) / s


def sample_tail(rgb_u8, depth_mm, labels):
    """One dataset item after the geometric transforms: (image (3,h,w) f32, depth (1,h,w) f32 metres, seg (1,h,w) int64)."""
    seg = torch.where(labels[None] > 0, 1, 0).type(torch.long)
    return normalize(to_tensor(rgb_u8)), depth_mm[None] / 1000.0, seg


def nested(tensor_list):
    """nested_tensor_from_tensor_list (src/util/misc.py:291-313): zero-pad to the largest (h,w); mask True = padding."""
    c = tensor_list[0].shape[0]
    h, w = max(t.shape[1] for t in tensor_list), max(t.shape[2] for t in tensor_list)
    out = torch.zeros((len(tensor_list), c, h, w), dtype=tensor_list[0].dtype)
    mask = torch.ones((len(tensor_list), h, w), dtype=torch.bool)
    for t, o, m in zip(tensor_list, out, mask):
        o[:, :t.shape[1], :t.shape[2]].copy_(t)
        m[:t.shape[1], :t.shape[2]] = False
    return out, mask


def collate(samples):
    """samples: list of (rgb_u8, depth_mm, labels) -> dict like engine batches (collate_fn_aux, src/util/misc.py:273-280)."""
    items = [sample_tail(*s) for s in samples]
    images, mask = nested([i[0] for i in items])
    depth, _ = nested([i[1] for i in items])
    seg, _ = nested([i[2] for i in items])
    return {"images": images, "pad_mask": mask, "depth": depth, "seg": seg}


def synth_samples(sizes, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for h, w in sizes:
        rgb = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8)
        depth = torch.randint(0, 12000, (h, w), generator=g, dtype=torch.int32)
        labels = torch.tensor([0, 0, 1, 2, 255], dtype=torch.uint8)[torch.randint(0, 5, (h, w), generator=g)]
        out.append((rgb, depth, labels))
    return out

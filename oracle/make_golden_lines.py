"""Generate tests/golden/line_transforms.npz with the REAL reference's target arithmetic on CPU.

TEST INFRASTRUCTURE ONLY.  Usage: python -m oracle.make_golden_lines.

What runs is the reference's own code, imported unmodified under oracle/ref_stubs.py:
  * crop / hflip / vflip / resize / Normalize of src/datasets/transforms_depth.py:59-128, 206-250, 316-372, 618-660 on line
    targets, polygon ids and polygon centres,
  * generate_line_labels + ConvertLinePolysToMask + the tail of DataLoadPreprocess.__getitem__ (depth / 1000, label > 0, the
    with_center concat, key removal) of src/datasets/glassrgbd_norhint.py:121-148, 161-193, 236-299 on decoded arrays written
    to temporary PNG / JSON files.
torchvision (absent, un-pinned) only moves PIXELS in these functions; its calls get stand-ins that carry no target arithmetic:
an image is an object with `.size` / `.shape`, F.crop / F.resize return one of the new size, F.hflip / F.vflip the same one.
For the item-assembly case the image really is decoded (Pillow is installed) and F.to_tensor / F.normalize are the restatements
of oracle/collate_ref.py (those two lines stay parity-unpinned as DESIGN.md section 2 records).  shapely is absent too: crop()
reaches it only for a polygon that keeps <= 3 of its lines, so the centre cases here keep > 3 lines of every polygon or drop the
polygon entirely; the shapely branch stays parity-unpinned."""
import json
import os
import tempfile
import types

import numpy as np
import torch

from . import collate_ref, ref_stubs

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class _Img:
    """Image stand-in: carries a size, no pixels."""

    def __init__(self, w, h):
        self.size = (int(w), int(h))
        self.shape = (3, int(h), int(w))
        self.width, self.height = int(w), int(h)


def _install_functional():
    import torchvision.transforms.functional as F      # the empty stand-in module of ref_stubs
    F.crop = lambda img, i, j, h, w: _Img(w, h)
    F.hflip = lambda img: img
    F.vflip = lambda img: img
    F.resize = lambda img, size, interpolation=None: _Img(size[1], size[0])
    F.normalize = lambda img, mean, std: img
    F.to_tensor = lambda img: img
    return F


def line_cases():
    """(name, w, h, lines (n,4) fp32, poly_ids, op, arg) - every clipping branch of crop(), both flips, resizes."""
    g = torch.Generator().manual_seed(97)
    cases = []
    w, h = 640, 480
    rnd = torch.rand((40, 4), generator=g) * torch.tensor([w, h, w, h], dtype=torch.float32)
    left_first = rnd[:, 0] > rnd[:, 2]
    rnd[left_first] = rnd[left_first][:, [2, 3, 0, 1]]
    edge = torch.tensor([[0., 0., 640., 480.], [100., 50., 100., 400.], [100., 400., 100., 50.], [50., 200., 600., 200.],
                         [120., 40., 121., 470.], [30., 60., 620., 61.], [319.5, 0., 320.5, 480.], [0., 239.5, 640., 240.5],
                         [200., 100., 200., 100.], [150., 150., 450., 150.], [150., 150., 150., 330.], [150., 330., 450., 330.],
                         [450., 150., 450., 330.], [149., 149., 451., 331.], [160., 90., 480., 390.], [480., 90., 160., 390.]])
    lines = torch.cat([rnd, edge], 0)
    ids = torch.arange(lines.shape[0]) // 4
    for k, region in enumerate([(90, 160, 300, 320), (150, 150, 180, 300), (0, 0, 480, 640), (37, 211, 401, 387), (200, 300, 64, 64),
                                (10, 5, 460, 630), (239, 319, 2, 2)]):
        cases.append(("crop%d" % k, w, h, lines, ids, "crop", region))
    cases.append(("hflip", w, h, lines, ids, "hflip", None))
    cases.append(("vflip", w, h, lines, ids, "vflip", None))
    for k, (size, mx) in enumerate([(480, 1024), (800, 1024), (1024, 1024), (400, None), (600, None), (690, 1024), ((512, 384), None)]):
        cases.append(("resize%d" % k, w, h, lines, ids, "resize", (size, mx)))
    cases.append(("resize_tall", 480, 640, lines[:, [1, 0, 3, 2]].contiguous(), ids, "resize", (788, 1024)))
    cases.append(("normalize", w, h, lines, ids, "normalize", None))
    # a chain as RandomSelect's second branch runs it: flip -> resize -> crop -> resize -> normalize
    cases.append(("chain", w, h, lines, ids, "chain", [("hflip", None), ("resize", (500, None)), ("crop", (20, 33, 390, 540)),
                                                         ("resize", (704, 1024)), ("normalize", None)]))
    cases.append(("chain_v", w, h, lines, ids, "chain", [("vflip", None), ("resize", (600, None)), ("crop", (101, 7, 384, 600)),
                                                           ("resize", (480, 1024)), ("normalize", None)]))
    return cases


def centre_cases():
    """Polygons (closed chains of lines) with centres: crop windows under which every polygon keeps > 3 lines or none."""
    polys = [[(100., 100.), (300., 90.), (320., 250.), (200., 330.), (90., 260.)],
             [(400., 300.), (600., 310.), (610., 400.), (500., 460.), (390., 420.), (380., 350.)],
             [(10., 10.), (60., 12.), (62., 50.), (30., 70.), (8., 48.)]]
    lines, ids, centres = [], [], []
    for pid, pts in enumerate(polys):
        n = len(pts)
        cx, cy = sum(p[0] for p in pts) / n, sum(p[1] for p in pts) / n
        for a in range(n):
            b = (a + 1) % n
            lines.append([pts[a][0], pts[a][1], pts[b][0], pts[b][1]])
            ids.append(pid + 3)
            centres.append([cx, cy])
    lines, ids, centres = torch.tensor(lines), torch.tensor(ids), torch.tensor(centres)
    out = []
    for k, (region, flipped) in enumerate([((60, 80, 400, 540), False), ((0, 0, 480, 640), False), ((80, 70, 300, 300), False),
                                           ((60, 20, 400, 550), True)]):
        out.append(("centre%d" % k, 640, 480, lines, ids, centres, region, flipped))
    return out


def run_op(T, img, target, op, arg):
    if op == "crop":
        img, target = T.crop(img, target, arg)
    elif op == "hflip":
        img, target = T.hflip(img, target)
    elif op == "vflip":
        img, target = T.vflip(img, target)
    elif op == "resize":
        img, target, _ = T.resize(img, target, arg[0], arg[1])
    elif op == "normalize":
        img, target = T.Normalize([0.538, 0.494, 0.453], [0.257, 0.263, 0.273])(img, target)
    return img, target


def item_case(tmp):
    """Decoded arrays -> files -> the reference's DataLoadPreprocess.__getitem__ (val chain without the resize)."""
    from PIL import Image
    rng = np.random.RandomState(5)
    h, w = 48, 64
    rgb = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    depth_mm = rng.randint(0, 9000, (h, w)).astype(np.uint16)
    depth_mm[rng.rand(h, w) < 0.1] = 0
    labels = (rng.rand(h, w) < 0.4).astype(np.uint8) * rng.randint(1, 4, (h, w)).astype(np.uint8)
    shapes = [{"poly_id": 7, "points": [[5.5, 4.0], [40.25, 6.0], [70.0, 30.5], [20.0, 52.0], [-3.0, 20.0]]},
              {"poly_id": 2, "points": []},
              {"poly_id": 9, "points": [[10.0, 10.0], [30.0, 12.5], [28.0, 40.0], [12.0, 38.0]]}]
    for sub in ("images", "depth", "seg", "json"):
        os.makedirs(os.path.join(tmp, sub))
    Image.fromarray(rgb).save(os.path.join(tmp, "images", "a.png"))
    Image.fromarray(depth_mm).save(os.path.join(tmp, "depth", "a.png"))
    Image.fromarray(labels).save(os.path.join(tmp, "seg", "a.png"))
    with open(os.path.join(tmp, "json", "a.json"), "w") as f:
        json.dump({"shapes": shapes, "imageWidth": w, "imageHeight": h, "imageId": 31}, f)
    with open(os.path.join(tmp, "list.txt"), "w") as f:
        f.write("a\n")
    with open(os.path.join(tmp, "ids.json"), "w") as f:
        json.dump({"images": [{"id": 31, "file_name": "a.png"}]}, f)
    return rgb, depth_mm, labels, shapes


def main():
    ref_stubs.install()
    F = _install_functional()
    import datasets.transforms_depth as T                      # /root/reference/src/datasets/transforms_depth.py
    out = {}
    names = []
    for name, w, h, lines, ids, op, arg in line_cases():
        target = {"lines": lines.clone(), "poly_ids": ids.clone(), "labels": torch.zeros(len(lines), dtype=torch.int64),
                  "area": torch.ones(len(lines)), "iscrowd": torch.zeros(len(lines))}
        img = _Img(w, h)
        for o, a in (arg if op == "chain" else [(op, arg)]):
            img, target = run_op(T, img, target, o, a)
        out[name + "_in"] = lines.numpy()
        out[name + "_ids_in"] = ids.numpy()
        out[name + "_out"] = target["lines"].numpy()
        out[name + "_ids_out"] = target["poly_ids"].numpy()
        out[name + "_size"] = np.array([w, h] + list(img.size), dtype=np.int64)
        names.append(name)
    for name, w, h, lines, ids, centres, region, flipped in centre_cases():
        target = {"lines": lines.clone(), "poly_ids": ids.clone(), "poly_centers": centres.clone(),
                  "labels": torch.zeros(len(lines), dtype=torch.int64), "area": torch.ones(len(lines)), "iscrowd": torch.zeros(len(lines))}
        img = _Img(w, h)
        if flipped:
            img, target = T.hflip(img, target)
            out[name + "_flipped_in"] = target["lines"].numpy().copy()
            out[name + "_flipped_centres"] = target["poly_centers"].numpy().copy()
        img, target = T.crop(img, target, region)
        img, target, _ = T.resize(img, target, 512, 1024)
        img, target = T.Normalize([0.5] * 3, [0.2] * 3)(img, target)
        out[name + "_in"], out[name + "_ids_in"], out[name + "_centres_in"] = lines.numpy(), ids.numpy(), centres.numpy()
        out[name + "_region"] = np.array(region, dtype=np.int64)
        out[name + "_out"], out[name + "_ids_out"] = target["lines"].numpy(), target["poly_ids"].numpy()
        out[name + "_centres_out"] = target["poly_centers"].numpy()
        names.append(name)

    # ---- item assembly through the reference's dataset class
    F.to_tensor = lambda img: collate_ref.to_tensor(torch.from_numpy(np.asarray(img).copy()))
    F.normalize = lambda t, mean, std: collate_ref.normalize(t, mean, std)
    import datasets.glassrgbd_norhint as D                      # /root/reference/src/datasets/glassrgbd_norhint.py
    with tempfile.TemporaryDirectory() as tmp:
        rgb, depth_mm, labels, shapes = item_case(tmp)
        args = types.SimpleNamespace(filenames_file_train=os.path.join(tmp, "list.txt"), filenames_file_eval=os.path.join(tmp, "list.txt"),
                                     glassrgbd_images_json=os.path.join(tmp, "ids.json"), data_path=os.path.join(tmp, "images"),
                                     gt_line_path=os.path.join(tmp, "json"), gt_depth_path=os.path.join(tmp, "depth"),
                                     gt_seg_path=os.path.join(tmp, "seg"), with_center=True)
        chain = T.Compose([T.ToTensor(), T.Normalize([0.538, 0.494, 0.453], [0.257, 0.263, 0.273])])
        ds = D.DataLoadPreprocess(args, "val", transforms=chain)
        image, depth_gt, seg_gt, targets, path = ds[0]
    out["item_rgb"], out["item_depth_mm"], out["item_labels"] = rgb, depth_mm.astype(np.int32), labels
    out["item_shapes"] = np.array(json.dumps(shapes))
    out["item_image"], out["item_depth"], out["item_seg"] = image.numpy(), depth_gt.numpy(), seg_gt.numpy()
    for k in ("lines", "labels", "poly_ids", "image_id", "orig_size", "size"):
        out["item_t_" + k] = targets[k].numpy()
    out["item_keys"] = np.array(sorted(targets.keys()))
    out["names"] = np.array(names)
    path = os.path.join(GOLDEN_DIR, "line_transforms.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024), len(names), "cases")


if __name__ == "__main__":
    main()

"""Generate tests/golden/collate.npz with the REAL reference's collate_fn_aux (src/util/misc.py:273-280) on CPU.

TEST INFRASTRUCTURE ONLY.  Usage: python -m oracle.make_golden_collate.  The per-sample tensors come from
oracle/collate_ref.sample_tail (torchvision, whose to_tensor / normalize the reference calls, is not installed); padding,
masks and batching are the reference's own code, imported unmodified under oracle/ref_stubs.py."""
import os

import numpy as np

from . import collate_ref, ref_stubs

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SIZES, SEED = [(20, 28), (24, 20), (16, 32)], 41


def main():
    ref_stubs.install()
    from util.misc import collate_fn_aux             # /root/reference/src/util/misc.py:273
    samples = collate_ref.synth_samples(SIZES, SEED)
    items = [collate_ref.sample_tail(*s) + ({"lines": None}, "img%d" % i) for i, s in enumerate(samples)]
    images, depth, seg, _, _ = collate_fn_aux(items)
    out = {"images": images.tensors.numpy(), "pad_mask": images.mask.numpy(), "depth": depth.tensors.numpy(),
           "depth_mask": depth.mask.numpy(), "seg": seg.tensors.numpy()}
    for i, (rgb, d, l) in enumerate(samples):
        out["rgb%d" % i], out["depth_mm%d" % i], out["labels%d" % i] = rgb.numpy(), d.numpy(), l.numpy()
    path = os.path.join(GOLDEN_DIR, "collate.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024), out["images"].shape, out["pad_mask"].sum())


if __name__ == "__main__":
    main()

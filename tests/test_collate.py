"""Device batch assembly (first slice): ToTensor + Normalize + depth/seg conversion + padding collate.
tests/golden/collate.npz: padding / masks / batching by the reference's OWN collate_fn_aux (oracle/make_golden_collate.py).
Everything here is exact fp32 / integer arithmetic: the bar is bit-exact."""
import os

import numpy as np
import pytest
import torch

from gw_depth_amd import hip
from gw_depth_amd.data import device_collate
from oracle import collate_ref
from tests.fake_device import FakeDevice

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collate.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLDEN))


def samples_of(gold):
    return [(torch.from_numpy(gold["rgb%d" % i]), torch.from_numpy(gold["depth_mm%d" % i]), torch.from_numpy(gold["labels%d" % i]))
            for i in range(3)]


def check(out, gold):
    np.testing.assert_array_equal(out["images"].float().cpu().numpy(), gold["images"])
    np.testing.assert_array_equal(out["pad_mask"].cpu().numpy(), gold["pad_mask"])
    np.testing.assert_array_equal(out["pad_mask"].cpu().numpy(), gold["depth_mask"])
    np.testing.assert_array_equal(out["depth"].cpu().numpy(), gold["depth"])
    np.testing.assert_array_equal(out["seg"].cpu().numpy(), gold["seg"])
    assert out["seg"].dtype == torch.int64 and out["depth"].dtype == torch.float32 and out["pad_mask"].dtype == torch.bool


def test_oracle_collate_matches_reference_collate(gold):
    check(collate_ref.collate(samples_of(gold)), gold)


@pytest.fixture()
def fake():
    hip.set_library(FakeDevice())
    yield
    hip.set_library(None)


def test_host_logic_matches_reference_collate(fake, gold):
    out = device_collate(samples_of(gold), device="cpu")
    check(out, gold)
    assert out["images"].permute(0, 2, 3, 1).is_contiguous()                  # the model's pixel-major read is in place
    with pytest.raises(ValueError):
        device_collate([], device="cpu")
    with pytest.raises(ValueError):
        device_collate([(torch.zeros(4, 4, 3, dtype=torch.uint8), torch.zeros(3, 4, dtype=torch.int32), None)], device="cpu")


@pytest.fixture()
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    hip.set_library(None)
    return hip.library()


@pytest.mark.gpu
def test_kernel_matches_reference_collate_bit_exact(dev, gold):
    check(device_collate(samples_of(gold), device="cuda"), gold)
    out = device_collate([(r, None, None) for r, _, _ in samples_of(gold)], device="cuda")      # images only
    np.testing.assert_array_equal(out["images"].cpu().numpy(), gold["images"])
    assert "depth" not in out and "seg" not in out


@pytest.mark.gpu
def test_kernel_full_size_batch_vs_oracle(dev):
    """8 frames around 480x640 (ragged): fp32 bit-exact against the oracle; bf16 = the rounded fp32 values; the batch
    feeds the model's pixel-major layout without a copy."""
    sizes = [(480, 640), (480, 640), (468, 640), (480, 620), (480, 640), (400, 500), (480, 640), (480, 640)]
    samples = collate_ref.synth_samples(sizes, seed=7)
    ref = collate_ref.collate(samples)
    out = device_collate(samples, device="cuda")
    for k in ("images", "pad_mask", "depth", "seg"):
        assert torch.equal(out[k].cpu(), ref[k]), k
    out16 = device_collate(samples, device="cuda", dtype=torch.bfloat16)
    assert torch.equal(out16["images"].cpu(), ref["images"].to(torch.bfloat16))
    assert out16["images"].permute(0, 2, 3, 1).is_contiguous()

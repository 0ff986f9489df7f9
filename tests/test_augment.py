"""Geometric input transforms on the device (second slice): flip / crop / resize of RGB, depth and label maps,
bit-exact with Pillow (the library the reference calls through torchvision, src/datasets/transforms_depth.py:59-372), and the matching
arithmetic on the line targets.  tests/golden/pil_resize.npz: inputs and outputs produced by Pillow (oracle/make_golden_pil_resize.py).
Everything here is integer / index work: the bar is bit-exact."""
import os

import numpy as np
import pytest
import torch

from gw_depth_amd import data, hip
from oracle import pil_resize_ref as ref
from tests.fake_device import FakeDevice

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pil_resize.npz")
N_CASES = 7


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLDEN))


def test_oracle_resizes_match_the_pillow_golden_vectors(gold):
    for n in range(N_CASES):
        oh, ow = (int(v) for v in gold[f"size{n}"])
        np.testing.assert_array_equal(ref.resize_bilinear_u8(gold[f"rgb{n}"], oh, ow), gold[f"rgb_out{n}"])
        np.testing.assert_array_equal(ref.resize_nearest(gold[f"dep{n}"], oh, ow), gold[f"dep_out{n}"])
        np.testing.assert_array_equal(ref.resize_nearest(gold[f"lab{n}"], oh, ow), gold[f"lab_out{n}"])
        flipped = gold[f"rgb{n}"][:, ::-1] if n % 2 == 0 else gold[f"rgb{n}"][::-1]
        np.testing.assert_array_equal(ref.resize_bilinear_u8(np.ascontiguousarray(flipped), oh, ow), gold[f"rgb_flip_out{n}"])


def test_oracle_matches_the_installed_pillow_on_dataset_like_sizes():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for (h, w, oh, ow) in [(270, 480, 200, 355), (135, 240, 300, 533), (240, 135, 426, 240)]:
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        np.testing.assert_array_equal(ref.resize_bilinear_u8(rgb, oh, ow), np.asarray(Image.fromarray(rgb).resize((ow, oh), Image.BILINEAR)))
        dep = rng.integers(0, 12000, (h, w)).astype(np.int32)
        np.testing.assert_array_equal(ref.resize_nearest(dep, oh, ow), np.asarray(Image.fromarray(dep, mode="I").resize((ow, oh), Image.NEAREST)))


def test_product_tables_equal_the_oracle_tables():
    rng = np.random.default_rng(5)
    for _ in range(150):
        a, b = int(rng.integers(2, 1500)), int(rng.integers(2, 1500))
        b1, k1 = data.bilinear_tables(a, b)
        b2, k2 = ref.bilinear_tables(a, b)
        np.testing.assert_array_equal(b1, b2)
        np.testing.assert_array_equal(k1, k2)
        np.testing.assert_array_equal(data.nearest_table(a, b), ref.nearest_table(a, b))


def _chain_reference(rgb, dep, lab, p):
    """The same chain through the oracle, step by step (each step a new image, as the reference's Compose does)."""
    if p["flip"] == "h":
        rgb, dep, lab = rgb[:, ::-1], dep[:, ::-1], lab[:, ::-1]
    if p["flip"] == "v":
        rgb, dep, lab = rgb[::-1], dep[::-1], lab[::-1]
    rgb, dep, lab = np.ascontiguousarray(rgb), np.ascontiguousarray(dep), np.ascontiguousarray(lab)
    for step in p["steps"]:
        h, w = rgb.shape[:2]
        if step[0] == "resize":
            oh, ow = data.resized_shape(w, h, step[1], step[2])
            rgb, dep, lab = ref.resize_bilinear_u8(rgb, oh, ow), ref.resize_nearest(dep, oh, ow), ref.resize_nearest(lab, oh, ow)
        else:
            i, j, ch, cw = step[1]
            rgb, dep, lab = (np.ascontiguousarray(a[i:i + ch, j:j + cw]) for a in (rgb, dep, lab))
    return rgb, dep, lab


PARAMS = [
    {"flip": None, "steps": [("resize", 96, 1024)]},
    {"flip": "h", "steps": [("resize", 120, 160)]},
    {"flip": "v", "steps": [("resize", 80, None), ("crop", (5, 9, 50, 61)), ("resize", 100, 1024)]},
    {"flip": "h", "steps": [("resize", 70, None), ("crop", (0, 0, 40, 40)), ("resize", 64, 1024)]},
    {"flip": "v", "steps": []},
]


def _run_chain(p, device):
    rng = np.random.default_rng(11)
    h, w = 72, 128
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    dep = rng.integers(0, 12000, (h, w)).astype(np.int32)
    lab = rng.integers(0, 3, (h, w)).astype(np.uint8)
    lines = torch.tensor([[10.0, 20.0, 100.0, 60.0], [5.0, 5.0, 5.0, 70.0], [0.0, 71.0, 127.0, 0.0], [120.0, 10.0, 126.0, 12.0]])
    out = data.DeviceAugment.apply(torch.from_numpy(rgb).to(device), torch.from_numpy(dep).to(device), torch.from_numpy(lab).to(device), lines, p)
    want = _chain_reference(rgb, dep, lab, p)
    for got, w_ in zip(out[:3], want):
        np.testing.assert_array_equal(got.cpu().numpy(), w_)
    return out


@pytest.mark.parametrize("p", PARAMS, ids=[str(i) for i in range(len(PARAMS))])
def test_device_augment_host_logic_on_the_cpu_stand_in(p):
    hip.set_library(FakeDevice())
    try:
        _run_chain(p, "cpu")
    finally:
        hip.set_library(None)


def test_line_target_transforms():
    lines = torch.tensor([[10.0, 20.0, 100.0, 60.0], [5.0, 5.0, 5.0, 70.0]])
    hf = data.hflip_lines(lines, 128)
    assert torch.equal(hf, torch.tensor([[28.0, 60.0, 118.0, 20.0], [123.0, 70.0, 123.0, 5.0]]))          # swapped end points, x -> w - x
    assert torch.equal(data.hflip_lines(hf, 128), lines)
    vf = data.vflip_lines(lines.clone(), 72)
    assert torch.equal(vf, torch.tensor([[10.0, 52.0, 100.0, 12.0], [5.0, 2.0, 5.0, 67.0]]))               # vertical line: upper point first again
    assert torch.equal(data.resize_lines(lines, 128, 72, 256, 144), lines * 2)
    # crop (i=10, j=20, h=40, w=60): line 0 enters at x=0 on its slope and leaves through the bottom edge; line 1 lies left of the window
    cl, keep = data.crop_lines(lines, (10, 20, 40, 60))
    assert keep.tolist() == [True, False]
    slope = 40.0 / 90.0
    x1, y1 = 0.0, 50.0 + (0.0 - 80.0) * slope            # shifted line: (-10, 10) -> (80, 50)
    # ... x2 = 60 first gives y2 = 41.1 > h = 40, so the line is cut again at the bottom edge: y2 = 40, x2 = x1 + (40 - y1) / slope
    assert torch.allclose(cl, torch.tensor([[x1, y1, (40.0 - y1) / slope, 40.0]]), atol=1e-4)
    assert data.resized_shape(1280, 720, 480, 1024) == (480, 853) and data.resized_shape(720, 1280, 800, 1024) == (1024, 576)
    assert data.resized_shape(640, 480, 480, 1024) == (480, 640) and data.resized_shape(100, 50, (30, 20)) == (20, 30)


def test_params_follow_the_reference_recipe():
    aug = data.DeviceAugment(train=True, seed=0)
    kinds = set()
    for _ in range(200):
        p = aug.params(1280, 720)
        assert p["flip"] in (None, "h", "v")
        kinds.add(tuple(s[0] for s in p["steps"]))
        for s in p["steps"]:
            if s[0] == "resize":
                assert s[1] in data.DeviceAugment.SCALES + [400, 500, 600]
            else:
                i, j, ch, cw = s[1]
                assert 384 <= ch <= 600 and 384 <= cw <= 600 and i >= 0 and j >= 0
    assert kinds == {("resize",), ("resize", "crop", "resize")}
    assert sorted(n for n, _ in p["jitter"]) == ["brightness", "contrast", "hue", "saturation"]
    assert data.DeviceAugment(train=False).params(1280, 720) == {"flip": None, "steps": [("resize", 1024, 1024)]}


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("p", PARAMS, ids=[str(i) for i in range(len(PARAMS))])
def test_device_augment_kernels_bit_exact(p):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    hip.set_library(None)
    _run_chain(p, "cuda")


@pytest.mark.gpu
def test_device_resizes_match_golden_and_pillow_at_dataset_size(gold):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    hip.set_library(None)
    for n in range(N_CASES):
        oh, ow = (int(v) for v in gold[f"size{n}"])
        rgb = torch.from_numpy(gold[f"rgb{n}"]).cuda()
        np.testing.assert_array_equal(data.device_resize_rgb(rgb, (oh, ow)).cpu().numpy(), gold[f"rgb_out{n}"])
        np.testing.assert_array_equal(data.device_resize_rgb(rgb, (oh, ow), hflip=n % 2 == 0, vflip=n % 2 == 1).cpu().numpy(), gold[f"rgb_flip_out{n}"])
        np.testing.assert_array_equal(data.device_resize_nearest(torch.from_numpy(gold[f"dep{n}"]).cuda(), (oh, ow)).cpu().numpy(), gold[f"dep_out{n}"])
        np.testing.assert_array_equal(data.device_resize_nearest(torch.from_numpy(gold[f"lab{n}"]).cuda(), (oh, ow)).cpu().numpy(), gold[f"lab_out{n}"])
    # full-size frame (the dataset's 720 x 1280) through the oracle; idempotence of the double flip as a size-independent property
    rng = np.random.default_rng(8)
    rgb = rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    t = torch.from_numpy(rgb).cuda()
    oh, ow = data.resized_shape(1280, 720, 480, 1024)
    got = data.device_resize_rgb(t, (oh, ow)).cpu().numpy()
    np.testing.assert_array_equal(got, ref.resize_bilinear_u8(rgb, oh, ow))
    twice = data.device_resize_rgb(data.device_resize_rgb(t, (720, 1280), hflip=True), (720, 1280), hflip=True)
    assert torch.equal(twice, t)


# ------------------------------------------------------------------------------------------------------------ colour jitter
COLOR_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pil_color.npz")


@pytest.fixture(scope="module")
def cgold():
    return dict(np.load(COLOR_GOLDEN))


def _color_cases(cgold):
    for f in cgold["factors"]:
        for name in ("brightness", "contrast", "saturation"):
            yield name, float(f), cgold["%s_%.2f" % (name, f)]
    for f in cgold["hue_factors"]:
        yield "hue", float(f), cgold["hue_%.2f" % f]


def test_colour_oracle_matches_the_pillow_golden_vectors(cgold):
    from oracle import pil_color_ref as C
    fn = {"brightness": C.adjust_brightness, "contrast": C.adjust_contrast, "saturation": C.adjust_saturation, "hue": C.adjust_hue}
    np.testing.assert_array_equal(C.rgb_to_hsv(cgold["rgb"]), cgold["hsv"])
    for name, f, want in _color_cases(cgold):
        np.testing.assert_array_equal(fn[name](cgold["rgb"], f), want, err_msg="%s %.2f" % (name, f))


def test_colour_oracle_matches_the_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageEnhance
    from oracle import pil_color_ref as C
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    im = Image.fromarray(img)
    for f in (0.61, 0.99, 1.0, 1.39):
        np.testing.assert_array_equal(C.adjust_brightness(img, f), np.asarray(ImageEnhance.Brightness(im).enhance(f)))
        np.testing.assert_array_equal(C.adjust_contrast(img, f), np.asarray(ImageEnhance.Contrast(im).enhance(f)))
        np.testing.assert_array_equal(C.adjust_saturation(img, f), np.asarray(ImageEnhance.Color(im).enhance(f)))
    hsv = np.asarray(im.convert("HSV"))
    np.testing.assert_array_equal(C.rgb_to_hsv(img), hsv)
    np.testing.assert_array_equal(C.hsv_to_rgb(hsv), np.asarray(Image.fromarray(hsv, "HSV").convert("RGB")))


def test_jitter_params_and_hue_shift():
    import random
    ops = data.jitter_params(random.Random(3))
    assert sorted(n for n, _ in ops) == ["brightness", "contrast", "hue", "saturation"]
    for n, f in ops:
        assert (-0.4 <= f <= 0.4) if n == "hue" else (0.6 <= f <= 1.4)
    assert data.hue_shift(0.1) == 25 and data.hue_shift(-0.3) == 180 and data.hue_shift(0.0) == 0 and data.hue_shift(-0.001) == 0


@pytest.mark.gpu
def test_device_colour_jitter_bit_exact(cgold):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import pil_color_ref as C
    hip.set_library(None)
    rgb = torch.from_numpy(cgold["rgb"]).cuda()
    for name, f, want in _color_cases(cgold):
        got = data.device_color_jitter(rgb, [(name, f)])
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg="%s %.2f" % (name, f))
    # a full jitter chain on a frame-sized image against the oracle, step by step
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (480, 853, 3), dtype=np.uint8)
    ops = [("contrast", 1.23), ("hue", -0.31), ("brightness", 0.77), ("saturation", 1.38)]
    want = img
    fn = {"brightness": C.adjust_brightness, "contrast": C.adjust_contrast, "saturation": C.adjust_saturation, "hue": C.adjust_hue}
    for name, f in ops:
        want = fn[name](want, f)
    np.testing.assert_array_equal(data.device_color_jitter(torch.from_numpy(img).cuda(), ops).cpu().numpy(), want)


# ------------------------------------------------------------------------------------------------------------------------------
# Target arithmetic against the reference's OWN crop / hflip / vflip / resize / Normalize and dataset item assembly
# (tests/golden/line_transforms.npz, made by oracle/make_golden_lines.py from /root/reference/src/datasets/transforms_depth.py and
# glassrgbd_norhint.py).  Host tensors only: runs without a GPU.
def _golden_lines():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "line_transforms.npz"))
    return z, [str(n) for n in z["names"]]


def _apply_op(lines, ids, w, h, op, arg):
    if op == "crop":
        lines, ids, _, _ = data.crop_targets(lines, ids, None, arg)
        return lines, ids, arg[3], arg[2]
    if op == "hflip":
        return data.hflip_lines(lines, w), ids, w, h
    if op == "vflip":
        return data.vflip_lines(lines, h), ids, w, h
    if op == "resize":
        oh, ow = data.resized_shape(w, h, arg[0], arg[1])
        return data.resize_lines(lines, w, h, ow, oh), ids, ow, oh
    assert op == "normalize"
    return data.normalize_lines(lines, w, h), ids, w, h


def test_line_targets_equal_the_references_transforms_bit_for_bit():
    from oracle import make_golden_lines as G            # the case table (inputs + operations); the expected values come from the fixture
    z, names = _golden_lines()
    seen = 0
    for name, w, h, lines, ids, op, arg in G.line_cases():
        assert name in names
        assert np.array_equal(z[name + "_in"], lines.numpy())
        cur, cid, cw, ch = lines.clone(), ids.clone(), w, h
        for o, a in (arg if op == "chain" else [(op, arg)]):
            cur, cid, cw, ch = _apply_op(cur, cid, cw, ch, o, a)
        want = z[name + "_out"]
        assert cur.shape == want.shape, name
        assert np.array_equal(cur.numpy(), want, equal_nan=True), (name, np.abs(cur.numpy() - want).max())
        assert np.array_equal(cid.numpy(), z[name + "_ids_out"]), name
        assert [w, h, cw, ch] == z[name + "_size"].tolist(), name
        seen += 1
    assert seen >= 18
    # every clipping rule fired somewhere: a crop that moves each of the four coordinates to each of its two edges
    moved = set()
    for k in range(7):
        i, j, hh, ww = G.line_cases()[k][6]
        out = z["crop%d_out" % k]
        for c, edge in ((0, 0.0), (0, float(ww)), (1, 0.0), (1, float(hh)), (2, 0.0), (2, float(ww)), (3, 0.0), (3, float(hh))):
            if (out[:, c] == edge).any():
                moved.add((c, edge > 0))
    assert len(moved) == 8


def test_polygon_centres_follow_the_references_crop():
    from oracle import make_golden_lines as G
    z, names = _golden_lines()
    for name, w, h, lines, ids, centres, region, flipped in G.centre_cases():
        cur, cid, cc = lines.clone(), ids.clone(), centres.clone()
        if flipped:
            cur = data.hflip_lines(cur, w)
            cc = cc * torch.as_tensor([-1.0, 1.0]) + torch.as_tensor([float(w), 0.0])
            assert np.array_equal(cur.numpy(), z[name + "_flipped_in"]) and np.array_equal(cc.numpy(), z[name + "_flipped_centres"])
        cur, cid, cc, _ = data.crop_targets(cur, cid, cc, region)
        oh, ow = data.resized_shape(region[3], region[2], 512, 1024)
        cc = cc * torch.as_tensor([float(ow) / float(region[3]), float(oh) / float(region[2])])
        cur = data.resize_lines(cur, region[3], region[2], ow, oh)
        cur, cc = data.normalize_lines(cur, ow, oh, cc)
        assert np.array_equal(cid.numpy(), z[name + "_ids_out"]), name
        assert np.array_equal(cur.numpy(), z[name + "_out"]), name
        assert np.array_equal(cc.numpy(), z[name + "_centres_out"]), (name, cc.numpy(), z[name + "_centres_out"])


def test_item_assembly_equals_the_references_dataset_item():
    """Decoded arrays + polygon JSON -> target dict: the reference's DataLoadPreprocess.__getitem__ (fixture) vs data.assemble_item;
    the pixel tail is device_collate's (checked against the same item on the CPU stand-in)."""
    import json
    z, _ = _golden_lines()
    shapes = json.loads(str(z["item_shapes"]))
    rgb, dmm, lab = torch.from_numpy(z["item_rgb"]), torch.from_numpy(z["item_depth_mm"]), torch.from_numpy(z["item_labels"])
    r2, d2, l2, target = data.assemble_item(rgb, dmm, lab, shapes, 31, with_center=True)
    assert sorted(target.keys()) == [str(k) for k in z["item_keys"]]
    for k in ("lines", "labels", "poly_ids", "image_id", "orig_size", "size"):
        want = z["item_t_" + k]
        assert target[k].dtype == torch.from_numpy(want).dtype, k
        assert np.array_equal(target[k].numpy(), want), (k, target[k], want)
    hip.set_library(FakeDevice())
    try:
        batch = data.device_collate([(r2, d2, l2)], device="cpu")
    finally:
        hip.set_library(None)
    assert np.array_equal(batch["images"][0].numpy(), z["item_image"])
    assert np.array_equal(batch["depth"][0].numpy(), z["item_depth"])
    assert np.array_equal(batch["seg"][0].numpy(), z["item_seg"])


def test_crop_of_a_polygon_that_keeps_three_lines_uses_the_window_intersection():
    """The shapely branch (parity-unpinned, see data.crop_targets): the centre is the vertex mean of window INTERSECT polygon."""
    sq = torch.tensor([[10.0, 10.0, 50.0, 10.0], [50.0, 10.0, 50.0, 50.0], [50.0, 50.0, 10.0, 50.0], [10.0, 50.0, 10.0, 10.0]])
    ids = torch.zeros(4, dtype=torch.int64)
    centres = torch.full((4, 2), 30.0)
    lines, pid, cc, keep = data.crop_targets(sq, ids, centres, (0, 30, 100, 100))       # the window cuts the square's left edge away
    assert keep.tolist() == [True, True, True, False] and pid.tolist() == [0, 0, 0]
    # intersection = [30, 50] x [10, 50], shifted by the window origin (30, 0): the vertex mean lies inside [0, 20] x [10, 50]
    assert ((cc[:, 0] > 0) & (cc[:, 0] < 20.0) & (cc[:, 1] > 10.0) & (cc[:, 1] < 50.0)).all()
    assert torch.equal(cc[0], cc[1]) and torch.equal(cc[1], cc[2])

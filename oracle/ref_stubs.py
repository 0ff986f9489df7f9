"""Import the reference's own Python (read-only at /root/reference) in THIS container.

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_golden.py to run the real reference on CPU and
emit golden vectors into tests/golden/.  Nothing here travels into the product path and
nothing here is available on the GPU box (where /root/reference does not exist).

The reference's hot-path files import a few third-party modules that are not installed in
this image and that contribute NO arithmetic to the path: tkinter, turtle,
timm.models.layers (initialisers + identity DropPath at drop_path=0), cv2, shapely, docopt.
Those get empty in-process stand-ins in ``sys.modules``.  torchvision contributes the
ResNet-50 topology; that one is restated in oracle/resnet50_v15.py from the published
torchvision definition and plugged in behind the ``torchvision`` name.
"""
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("GWDEPTH_REFERENCE", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "src", "models"))


def install():
    """Register stand-in modules and put the reference on sys.path. Idempotent."""
    if "_gwdepth_ref_stubs" in sys.modules:
        return
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    _mod("_gwdepth_ref_stubs")
    from . import resnet50_v15 as rn

    # --- no-arithmetic stand-ins -------------------------------------------------------
    tk = _mod("tkinter")
    tk.messagebox = _mod("tkinter.messagebox", NO="no")
    _mod("turtle", forward=lambda *a, **k: None, color=lambda *a, **k: None)
    _mod("docopt", docopt=lambda *a, **k: {})
    cv2 = _mod("cv2", imwrite=lambda *a, **k: True, line=lambda *a, **k: None)
    cv2.__version__ = "0.0-stub"
    sh = _mod("shapely")
    sh.geometry = _mod("shapely.geometry", Polygon=object, mapping=lambda *a, **k: {})

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    class DropPath(torch.nn.Module):  # reference only ever builds it with drop_path > 0 (never)
        def __init__(self, p=0.0):
            super().__init__()
            assert p == 0.0

        def forward(self, x):
            return x

    timm = _mod("timm")
    timm.models = _mod("timm.models")
    timm.models.layers = _mod("timm.models.layers", DropPath=DropPath, to_2tuple=to_2tuple,
                              trunc_normal_=torch.nn.init.trunc_normal_)

    # --- torchvision: ResNet-50 topology restated in oracle/resnet50_v15.py ----------------
    tv = _mod("torchvision", __version__="0.15.0", _is_tracing=lambda: False)
    tv.models = _mod("torchvision.models", resnet50=rn.resnet50)
    tv.models._utils = _mod("torchvision.models._utils",
                            IntermediateLayerGetter=rn.IntermediateLayerGetter)
    tv.ops = _mod("torchvision.ops")
    tv.ops.misc = _mod("torchvision.ops.misc", interpolate=torch.nn.functional.interpolate)
    tv.datasets = _mod("torchvision.datasets", CocoDetection=object)

    class _Normalize(torch.nn.Module):
        def __init__(self, mean=None, std=None, *a, **k):
            super().__init__()

    tv.transforms = _mod("torchvision.transforms", Normalize=_Normalize,
                         InterpolationMode=types.SimpleNamespace(NEAREST=0, BILINEAR=2))
    tv.transforms.functional = _mod("torchvision.transforms.functional")

    for p in (os.path.join(REF_ROOT, "src"), REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)


def reference_args(extra=()):
    """argparse namespace exactly as src/main_glassrgbd.py builds it, with the one flag
    combination that constructs (SURVEY.md header) and --device cpu."""
    install()
    import argparse
    from args import get_args_parser  # /root/reference/src/args.py:4
    parser = argparse.ArgumentParser(parents=[get_args_parser()])
    argv = ["--with_line", "--with_center", "--with_dense", "--num_queries", "100",
            "--log_depth_error", "--device", "cpu", "--dropout", "0.0"] + list(extra)
    return parser.parse_args(argv)

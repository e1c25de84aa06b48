"""Container-only helper: make the reference's pure-torch NeRF path importable.

Used ONLY by tests/golden/make_golden.py (fixture generation).
Nothing here is imported by the product, the gpu tests, smoke() or bench.py.

The reference (vendored nerfstudio 0.3.3) imports many packages that are not in
this image and are irrelevant to the ray-sampling -> hash-grid -> MLP -> render
path.  We register permissive stub modules for them.
"""
import importlib
import os
import sys
import types

REF_ROOT = "/root/reference/nerfstudio-0.3.3"

_STUBS = [
    "jaxtyping", "nerfacc", "IPython", "wandb", "cv2", "tyro", "tyro.conf", "tyro.extras",
    "torchmetrics", "torchmetrics.functional", "torchmetrics.image", "torchmetrics.image.lpip",
    "tensorboard", "torch.utils.tensorboard", "viser", "viser.infra", "viser.transforms", "mediapy",
    "pyquaternion", "nuscenes", "nuscenes.nuscenes", "open3d", "msgpack_numpy", "socketio",
    "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "typeguard",
    "timm", "nuscenes.utils", "nuscenes.utils.splits", "nuscenes.utils.data_classes", "plotly",
    "plotly.graph_objects", "msgpack", "imageio", "appdirs", "gdown", "xatlas", "trimesh", "pymeshlab",
    "rawpy", "h5py", "pyngrok", "nuscenes.map_expansion", "nuscenes.map_expansion.map_api",
    "nuscenes.eval", "nuscenes.eval.common", "nuscenes.eval.common.utils", "shapely", "shapely.geometry",
    "descartes",
]


class _Dummy:
    """Callable / subscriptable / attribute-able placeholder."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        # used as a decorator -> return the function unchanged
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Dummy()

    def __getitem__(self, item):
        return _Dummy

    def __class_getitem__(cls, item):
        return cls

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()

    def __mro_entries__(self, bases):
        return (object,)

    def __or__(self, other):
        return _Dummy

    def __ror__(self, other):
        return _Dummy


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "nerfstudio"))


def install():
    """Insert stubs + put the reference on sys.path.  Idempotent."""
    if not reference_available():
        raise RuntimeError("reference tree not present (this only works in the build container)")
    sys.dont_write_bytecode = True
    for name in _STUBS:
        try:
            importlib.import_module(name)
            continue
        except Exception:
            pass
        if name not in sys.modules:
            sys.modules[name] = _StubModule(name)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def ref_modules():
    """Return a namespace with the reference modules used to make fixtures."""
    install()
    ns = types.SimpleNamespace()
    ns.encodings = importlib.import_module("nerfstudio.field_components.encodings")
    ns.mlp = importlib.import_module("nerfstudio.field_components.mlp")
    ns.spatial = importlib.import_module("nerfstudio.field_components.spatial_distortions")
    ns.activations = importlib.import_module("nerfstudio.field_components.activations")
    ns.rays = importlib.import_module("nerfstudio.cameras.rays")
    ns.cameras = importlib.import_module("nerfstudio.cameras.cameras")
    ns.samplers = importlib.import_module("nerfstudio.model_components.ray_samplers")
    ns.renderers = importlib.import_module("nerfstudio.model_components.renderers")
    ns.losses = importlib.import_module("nerfstudio.model_components.losses")
    ns.ps_losses = importlib.import_module("nerfstudio.model_components.PreSight.losses")
    ns.colliders = importlib.import_module("nerfstudio.model_components.scene_colliders")
    ns.ingp = importlib.import_module("nerfstudio.fields.PreSight.ingp_field")
    ns.ingp_ms = importlib.import_module("nerfstudio.fields.PreSight.ingp_field_ms")
    ns.prop = importlib.import_module("nerfstudio.fields.PreSight.prop_density_field")
    ns.prop_ms = importlib.import_module("nerfstudio.fields.PreSight.prop_density_field_ms")
    ns.sky = importlib.import_module("nerfstudio.fields.PreSight.sky_field")
    ns.sky_ms = importlib.import_module("nerfstudio.fields.PreSight.sky_field_ms")
    ns.field_heads = importlib.import_module("nerfstudio.field_components.field_heads")
    ns.math = importlib.import_module("nerfstudio.utils.math")
    return ns

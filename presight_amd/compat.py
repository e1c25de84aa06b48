"""Make the reference's dotted module paths resolve to this package, so that PreSight's YAML configs
(`!!python/object:nerfstudio.models.PreSight.nerfacto_nusc_ms.NerfactoNuscMSModelConfig`, see
nerfstudio-0.3.3/tests/data/configs/test_config1.yml) and checkpoints (state-dict keys) load unmodified for the
hot-path classes.  Only the hot-path modules are aliased; the rest of nerfstudio (data managers, trainer, viewer) is out
of scope and, if installed, is left untouched."""
from __future__ import annotations

import importlib
import sys
import types

_ALIASES = {
    "nerfstudio.field_components.encodings": "presight_amd.components",
    "nerfstudio.field_components.mlp": "presight_amd.components",
    "nerfstudio.field_components.embedding": "presight_amd.components",
    "nerfstudio.field_components.spatial_distortions": "presight_amd.components",
    "nerfstudio.field_components.activations": "presight_amd.components",
    "nerfstudio.field_components.field_heads": "presight_amd.fields",
    "nerfstudio.fields.PreSight.ingp_field": "presight_amd.fields",
    "nerfstudio.fields.PreSight.ingp_field_ms": "presight_amd.fields",
    "nerfstudio.fields.PreSight.prop_density_field": "presight_amd.fields",
    "nerfstudio.fields.PreSight.prop_density_field_ms": "presight_amd.fields",
    "nerfstudio.fields.PreSight.sky_field": "presight_amd.fields",
    "nerfstudio.fields.PreSight.sky_field_ms": "presight_amd.fields",
    "nerfstudio.cameras.rays": "presight_amd.rays",
    "nerfstudio.model_components.ray_samplers": "presight_amd.samplers",
    "nerfstudio.model_components.renderers": "presight_amd.renderers",
    "nerfstudio.model_components.scene_colliders": "presight_amd.renderers",
    "nerfstudio.model_components.losses": "presight_amd.losses",
    "nerfstudio.model_components.PreSight.losses": "presight_amd.losses",
    "nerfstudio.models.PreSight.nerfacto_nusc_ms": "presight_amd.model",
    "nerfstudio.engine.callbacks": "presight_amd.callbacks",
}


def install() -> None:
    """Register the aliases (idempotent).  The PreSight hot-path module paths are always redirected here; parent
    packages are only created when no real `nerfstudio` package is importable."""
    for name, target in _ALIASES.items():
        parts = name.split(".")
        for i in range(1, len(parts)):
            pkg = ".".join(parts[:i])
            if pkg not in sys.modules:
                m = types.ModuleType(pkg)
                m.__path__ = []
                sys.modules[pkg] = m
        mod = importlib.import_module(target)
        sys.modules[name] = mod
        setattr(sys.modules[".".join(parts[:-1])], parts[-1], mod)

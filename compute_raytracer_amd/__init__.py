"""rt355 -- MI355X-native stand-in for the reference's src/rendering-raycast path.

Host-side mirror (Python) of the reference's TypeScript classes for this path:

    Camera              <- src/rendering-raycast/camera.ts
    Light               <- src/rendering-raycast/light.ts
    Sphere              <- src/rendering-raycast/model/sphere.ts
    SceneRaytracing     <- src/rendering-raycast/scene-raytracing.ts   (sphere scenes)
    CubemapMaterial     <- src/material/cubemap-material.ts
    RendererRaytracing  <- src/rendering-raycast/renderer-raytracing.ts (the drop-in boundary)

All device work goes through the C-ABI library ``librt355.so`` (include/rt355.h);
there is no CPU fallback: constructing a renderer without the built HIP library
raises.
"""
from .camera import Camera
from .light import Light
from .sphere import Sphere
from .scene_raytracing import SceneRaytracing, synthetic_scene, BASELINE_CONFIGS
from .cubemap_material import CubemapMaterial
from .material import Material
from .mesh import Mesh
from .model import Model, Triangle
from .acceleration import AABB, BLAS, BVH, Node
from .renderer_raytracing import RendererRaytracing
from . import abi, tiles

__all__ = [
    "Camera", "Light", "Sphere", "SceneRaytracing", "synthetic_scene", "BASELINE_CONFIGS",
    "CubemapMaterial", "Material", "Mesh", "Model", "Triangle", "AABB", "BLAS", "BVH", "Node",
    "RendererRaytracing", "abi", "tiles",
]

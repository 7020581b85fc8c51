"""rt355 -- MI355X-native stand-in for the reference's src/rendering-raycast path.

Host-side mirror (Python) of the reference's TypeScript classes for this path:

    Camera              <- src/rendering-raycast/camera.ts
    Light               <- src/rendering-raycast/light.ts
    Sphere              <- src/rendering-raycast/model/sphere.ts
    SceneRaytracing     <- src/rendering-raycast/scene-raytracing.ts   (sphere scenes; triangle scenes as upload buffers)
    TriangleSoup, parse_obj, build_tree, Instances
                        -- the triangle path's host data as arrays (soup.py, acceleration/bvh.py, instances.py):
                           what obj-reader.ts, bvh.ts, model.ts and blas.ts produce, not how they hold it
    CubemapMaterial     <- src/material/cubemap-material.ts
    RendererRaytracing  <- src/rendering-raycast/renderer-raytracing.ts (the drop-in boundary)

All device work goes through the C-ABI library ``librt355.so`` (include/rt355.h);
there is no CPU fallback: constructing a renderer without the built HIP library
raises.
"""
from .camera import Camera
from .light import Light
from .sphere import Sphere
from .scene_raytracing import SceneRaytracing, TriMesh, load_mesh, load_mesh_file, synthetic_scene, BASELINE_CONFIGS
from .cubemap_material import CubemapMaterial
from .material import Material
from .soup import TriangleSoup, parse_obj
from .acceleration import MeshTree, build_tree
from .instances import Instances
from .renderer_raytracing import RendererRaytracing
from . import abi, tiles

__all__ = [
    "Camera", "Light", "Sphere", "SceneRaytracing", "synthetic_scene", "BASELINE_CONFIGS",
    "CubemapMaterial", "Material", "TriMesh", "load_mesh", "load_mesh_file", "TriangleSoup", "parse_obj",
    "MeshTree", "build_tree", "Instances",
    "RendererRaytracing", "abi", "tiles",
]

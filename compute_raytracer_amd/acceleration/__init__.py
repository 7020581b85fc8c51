"""Bottom-level trees of the triangle path (the RESULT of src/rendering-raycast/acceleration/bvh.ts, as arrays)."""
from .bvh import MeshTree, build_tree

__all__ = ["MeshTree", "build_tree"]

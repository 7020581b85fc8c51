"""Mirror of src/rendering-raycast/acceleration/{aabb,blas,bvh,node}.ts."""
from .node import Node
from .aabb import AABB
from .blas import BLAS
from .bvh import BVH

__all__ = ["Node", "AABB", "BLAS", "BVH"]

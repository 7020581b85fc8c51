"""Mirror of src/rendering-raycast/model/{triangle,model}.ts (Sphere lives in ../sphere.py)."""
from .triangle import Triangle
from .model import Model

__all__ = ["Triangle", "Model"]

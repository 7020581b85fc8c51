from .obj_reader import ObjectReader

__all__ = ["ObjectReader"]

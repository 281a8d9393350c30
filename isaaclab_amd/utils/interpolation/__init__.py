from .linear_interpolation import LinearInterpolation

__all__ = ["LinearInterpolation"]

from .interactive_scene import InteractiveScene

__all__ = ["InteractiveScene"]

"""Utilities of the MDP hot path (mirror of ``isaaclab.utils``)."""
from .buffers import CircularBuffer, DelayBuffer, TimestampedBuffer
from .configclass import configclass
from .interpolation import LinearInterpolation
from .timer import Timer
from .types import ArticulationActions

__all__ = ["CircularBuffer", "DelayBuffer", "TimestampedBuffer", "configclass", "LinearInterpolation", "Timer",
           "ArticulationActions"]

from .circular_buffer import CircularBuffer
from .delay_buffer import DelayBuffer
from .timestamped_buffer import TimestampedBuffer

__all__ = ["CircularBuffer", "DelayBuffer", "TimestampedBuffer"]

"""Stop-watch with the interface of ``LAB/utils/timer.py:21-171`` (``Timer`` as context manager / decorator / explicit
``start`` - ``stop``, the class-level ``timing_info`` registry, ``TimerError``).

Kernel launches are asynchronous, so a host clock around them measures the enqueue, not the work: with
``sync_device=True`` the watch drains the device's HIP streams when it starts and before it stops, which is the only way a
wall-clock figure says anything about this package's launches.  (Per-kernel numbers come from HIP events / rocprofv3, see
``bench.py``; this class is the coarse tool the reference uses around scene creation and terrain generation.)
"""
from __future__ import annotations

import time
from contextlib import ContextDecorator
from typing import ClassVar


class TimerError(Exception):
    """Misuse of a :class:`Timer` (started twice, stopped or queried while idle, unknown registry name)."""


def _drain_device() -> None:
    import torch

    if torch.cuda.is_available():
        torch.cuda.synchronize()


class Timer(ContextDecorator):
    """``with Timer("msg"): ...`` prints the span, ``Timer(name="key")`` files it under ``Timer.timing_info["key"]``."""

    timing_info: ClassVar[dict[str, float]] = {}

    def __init__(self, msg: str | None = None, name: str | None = None, sync_device: bool = False):
        self._msg, self._name, self._sync = msg, name, bool(sync_device)
        self._t0_ns: int | None = None     # None = idle
        self._span_s: float | None = None  # last completed start..stop span

    # -- state -------------------------------------------------------------------------------------
    def _running(self) -> bool:
        return self._t0_ns is not None

    @property
    def time_elapsed(self) -> float:
        """Seconds since ``start()`` (the watch keeps running)."""
        if not self._running():
            raise TimerError("Timer is not running. Use .start() to start it")
        return (time.perf_counter_ns() - self._t0_ns) * 1e-9

    @property
    def total_run_time(self) -> float:
        """Seconds between the last ``start()`` / ``stop()`` pair."""
        return self._span_s

    def __str__(self) -> str:
        return f"{self.time_elapsed:0.6f} seconds"

    # -- control -----------------------------------------------------------------------------------
    def start(self):
        if self._running():
            raise TimerError("Timer is running. Use .stop() to stop it")
        if self._sync:
            _drain_device()
        self._t0_ns = time.perf_counter_ns()

    def stop(self):
        if not self._running():
            raise TimerError("Timer is not running. Use .start() to start it")
        if self._sync:
            _drain_device()
        self._span_s = (time.perf_counter_ns() - self._t0_ns) * 1e-9
        self._t0_ns = None
        if self._name:
            Timer.timing_info[self._name] = self._span_s

    def __enter__(self) -> "Timer":
        self.start()
        return self

    def __exit__(self, exc_type, exc, tb):
        self.stop()
        if self._msg is not None:
            print(self._msg, f": {self._span_s:0.6f} seconds")
        return False

    @staticmethod
    def get_timer_info(name: str) -> float:
        try:
            return Timer.timing_info[name]
        except KeyError:
            raise TimerError(f"Timer {name} does not exist") from None

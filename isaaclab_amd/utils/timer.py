"""Wall-clock timer (``LAB/utils/timer.py:21-171``): context manager / decorator with a global registry."""
from __future__ import annotations

import time
from contextlib import ContextDecorator
from typing import Any, ClassVar


class TimerError(Exception):
    pass


class Timer(ContextDecorator):
    timing_info: ClassVar[dict[str, float]] = dict()

    def __init__(self, msg: str | None = None, name: str | None = None):
        self._msg = msg
        self._name = name
        self._start_time = None
        self._stop_time = None
        self._elapsed_time = None

    def __str__(self) -> str:
        return f"{self.time_elapsed:0.6f} seconds"

    @property
    def time_elapsed(self) -> float:
        return time.perf_counter() - self._start_time

    @property
    def total_run_time(self) -> float:
        return self._elapsed_time

    def start(self):
        if self._start_time is not None:
            raise TimerError("Timer is running. Use .stop() to stop it")
        self._start_time = time.perf_counter()

    def stop(self):
        if self._start_time is None:
            raise TimerError("Timer is not running. Use .start() to start it")
        self._stop_time = time.perf_counter()
        self._elapsed_time = self._stop_time - self._start_time
        self._start_time = None
        if self._name:
            Timer.timing_info[self._name] = self._elapsed_time

    def __enter__(self) -> "Timer":
        self.start()
        return self

    def __exit__(self, *exc_info: Any):
        self.stop()
        if self._msg is not None:
            print(self._msg, f": {self._elapsed_time:0.6f} seconds")

    @staticmethod
    def get_timer_info(name: str) -> float:
        if name not in Timer.timing_info:
            raise TimerError(f"Timer {name} does not exist")
        return Timer.timing_info.get(name)

"""Velocity-tracking locomotion task (``TASKS/manager_based/locomotion/velocity``)."""

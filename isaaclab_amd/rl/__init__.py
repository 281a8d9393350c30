"""What sits right behind the observation on the training side."""
from .running_standard_scaler import RunningStandardScaler

__all__ = ["RunningStandardScaler"]

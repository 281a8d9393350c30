"""State containers fed by a (synthetic / replayed) physics view (mirror of ``isaaclab.assets``)."""
from .articulation import Articulation, ArticulationCfg
from .articulation_data import ArticulationData
from .physics_view import SyntheticPhysicsView
from .rigid_object import RigidObject, RigidObjectData

__all__ = ["Articulation", "ArticulationCfg", "ArticulationData", "SyntheticPhysicsView", "RigidObject",
           "RigidObjectData"]

"""Synthetic terrain input generator (a restatement sized like ``ROUGH_TERRAINS_CFG``)."""
from .synthetic import SyntheticTerrain, SyntheticTerrainCfg, TerrainGeneratorCfg, convert_height_field_to_mesh, make_plane

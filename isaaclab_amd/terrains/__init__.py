"""Terrain input generators: the restated ``TerrainGenerator`` / ``ROUGH_TERRAINS_CFG`` pipeline
(height-field and box-built sub-terrains, layout, border, env origins) and the round-1 look-alike ``SyntheticTerrain``."""
from .synthetic import SyntheticTerrain, SyntheticTerrainCfg, convert_height_field_to_mesh, make_plane
from .synthetic import TerrainGeneratorCfg as SyntheticTerrainGeneratorCfg
from .terrain_generator import ROUGH_TERRAINS_CFG, GeneratedTerrain, TerrainGenerator, TerrainGeneratorCfg, TerrainImporterCfg

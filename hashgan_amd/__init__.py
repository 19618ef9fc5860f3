"""hashgan_amd -- MI355X-native retrieval evaluation (Hamming ranking, top-R, mAP)
for HashGAN's lib/metric.py.  See DESIGN.md."""
from .metric import MAPs, MAP, calc_map, MAP_per_query, RetrievalEngine, pack_codes, pack_labels, release_engines, pool_stats  # noqa: F401

__all__ = ["MAPs", "MAP", "calc_map", "MAP_per_query", "RetrievalEngine", "pack_codes", "pack_labels", "release_engines", "pool_stats"]

"""PointNet++ operator layer: same public names as the reference's ``model/pointnet2`` package."""
from . import fused_mlp  # noqa: F401  (first: fused_fp / fused_heads import their helpers from it, it re-exports their names)

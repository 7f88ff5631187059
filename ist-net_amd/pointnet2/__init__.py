"""PointNet++ operator layer: same public names as the reference's ``model/pointnet2`` package."""

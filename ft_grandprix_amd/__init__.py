"""ft_grandprix_amd -- MI355X-native batched racing-sim hot path (integrate + LiDAR + lap progress)."""
__version__ = "0.1.0"

"""Example drivers for the plugin surface ``Driver().process_lidar(ranges[, state]) -> (speed, steering_angle)``
(drivers/template.py of the reference).  They are this package's own; the reference's bundled drivers (ft_grandprix.nidc,
ft_grandprix.fast) run unmodified through the same surface when they are importable."""

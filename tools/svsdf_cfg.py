"""Point count of a named BASELINE config without importing the package (tools only)."""
POINTS = {"C1": 10_000, "C2": 100_000, "C3": 1_000_000, "C4": 4_000_000, "C5": 1_000_000, "NS": 1_000_000}


def default_points(cfg):
    return POINTS[cfg]

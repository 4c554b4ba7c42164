"""Angle handling without a hard astropy dependency: arguments may be plain
numbers (in the unit the reference documents) or astropy Quantities."""
import numpy as np

_TO_DEG = {"deg": 1.0, "arcmin": 1.0 / 60.0, "arcsec": 1.0 / 3600.0, "rad": 180.0 / np.pi}


def angle_value(q, unit, default_unit):
    """Numeric value of ``q`` in ``unit``; bare numbers are taken to be in ``default_unit``."""
    if hasattr(q, "to"):                       # astropy Quantity
        return float(q.to(unit).value)
    return float(q) * _TO_DEG[default_unit] / _TO_DEG[unit]

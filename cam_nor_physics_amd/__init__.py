"""Importable alias of the package directory `cam-nor-physics_amd/` (a hyphen cannot be imported).

All code lives in `cam-nor-physics_amd/`; this module only points the package search path there.
"""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cam-nor-physics_amd"))

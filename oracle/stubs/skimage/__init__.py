"""empty stand-in."""
from . import io, measure  # noqa: F401

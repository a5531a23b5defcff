"""empty stand-in."""

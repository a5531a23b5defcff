"""MI355X-native (gfx950) forward path behind visualDet3D's detector / operator API."""
__version__ = '0.1.0'

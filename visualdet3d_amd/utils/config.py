"""Config plumbing at the boundary: the reference hands detectors an ``easydict.EasyDict``
(``cfg.detector``; ``visualDet3D/utils/utils.py:137-152`` ``cfg_from_file``).  Any mapping with attribute
access works here; this class is the stand-in used when ``easydict`` is not installed."""
import importlib.util
import os
import shutil
import sys
import tempfile


class EasyDict(dict):
    """dict with attribute access, recursive on nested dicts (same surface as easydict.EasyDict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {})
        d.update(kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def cfg_from_file(cfg_filename: str):
    """Import a python config file and return its module-level ``cfg``
    (same contract as visualDet3D/utils/utils.py:137-152: the file is copied to a temp dir and imported)."""
    assert cfg_filename.endswith('.py')
    with tempfile.TemporaryDirectory() as temp_config_dir:
        temp_config_file = tempfile.NamedTemporaryFile(dir=temp_config_dir, suffix='.py')
        temp_config_name = os.path.basename(temp_config_file.name)
        shutil.copyfile(cfg_filename, os.path.join(temp_config_dir, temp_config_name))
        temp_module_name = os.path.splitext(temp_config_name)[0]
        sys.path.insert(0, temp_config_dir)
        try:
            spec = importlib.util.spec_from_file_location(temp_module_name, os.path.join(temp_config_dir, temp_config_name))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            cfg = getattr(mod, 'cfg')
            assert isinstance(cfg, dict)
        finally:
            sys.path.pop(0)
        temp_config_file.close()
    return cfg

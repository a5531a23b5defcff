"""String -> object registries, same contract as the reference's ``networks/utils/registry.py:2-50``:
``REG.register_module`` as a bare decorator, ``REG[name]`` lookup, duplicate names raise ``KeyError`` unless
``_register_module(obj, force=True)``."""
import inspect


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._module_dict = dict()

    def __repr__(self):
        return '{}(name={}, items={})'.format(self.__class__.__name__, self._name, list(self._module_dict.keys()))

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def __getitem__(self, key):
        return self._module_dict[key]

    def __contains__(self, key):
        return key in self._module_dict

    def _register_module(self, module_class, force=False):
        if not (inspect.isclass(module_class) or inspect.isfunction(module_class)):
            raise TypeError('module must be a class or function, but got {}'.format(type(module_class)))
        module_name = module_class.__name__
        if not force and module_name in self._module_dict:
            raise KeyError('{} is already registered in {}'.format(module_name, self.name))
        self._module_dict[module_name] = module_class

    def register_module(self, cls=None):
        self._register_module(cls)
        return cls


DATASET_DICT = Registry("datasets")
BACKBONE_DICT = Registry("backbones")
DETECTOR_DICT = Registry("detectors")
PIPELINE_DICT = Registry("pipelines")
AUGMENTATION_DICT = Registry("augmentation")
SAMPLER_DICT = Registry("sampler")

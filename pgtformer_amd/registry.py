"""Architecture registry with BasicSR's `register()/get()` surface.

The reference registers its archs in `basicsr.utils.registry.ARCH_REGISTRY`
(reference: archs/tdcrqvae3_arch.py:710; the decorator on PGTFormer is commented out,
archs/pgtformer_arch.py:489) and option files select them by `network_g.type`.  When basicsr is
importable the same registry object is used so a BasicSR pipeline finds these classes by name;
otherwise a built-in registry with the same surface is used.
"""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        # re-registration (module reload) replaces the entry instead of asserting
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map

    def keys(self):
        return self._obj_map.keys()


try:  # pragma: no cover - basicsr is absent in the build image
    from basicsr.utils.registry import ARCH_REGISTRY as _BASICSR

    class _Bridge(Registry):
        def _do_register(self, name, obj):
            super()._do_register(name, obj)
            try:
                _BASICSR._obj_map.pop(name, None)
                _BASICSR.register(obj)
            except Exception:
                pass

    ARCH_REGISTRY = _Bridge("arch")
except Exception:
    ARCH_REGISTRY = Registry("arch")


def build_network(opt):
    """BasicSR-style: opt is a `network_g` dict with a `type` key."""
    opt = dict(opt)
    return ARCH_REGISTRY.get(opt.pop("type"))(**opt)

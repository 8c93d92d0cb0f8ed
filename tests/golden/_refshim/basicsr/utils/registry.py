class _Registry:
    def __init__(self, name):
        self.name, self._m = name, {}
    def register(self, obj=None):
        def deco(o):
            self._m[o.__name__] = o
            return o
        return deco if obj is None else deco(obj)
    def get(self, name):
        return self._m[name]
ARCH_REGISTRY = _Registry("arch")

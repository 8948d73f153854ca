"""Name -> class registries; the plugin surface `basicsr.utils.registry` exposes
(reference `basicsr/utils/registry.py:4-82`: register as decorator or call, KeyError on an
unknown name, AssertionError on a duplicate)."""


class Registry:
    def __init__(self, name):
        self._name = name
        self._table = {}

    def register(self, obj=None):
        def _add(o):
            key = o.__name__
            assert key not in self._table, (f"An object named '{key}' was already registered "
                                            f"in '{self._name}' registry!")
            self._table[key] = o
            return o
        if obj is None:
            return _add
        _add(obj)

    def get(self, name):
        try:
            return self._table[name]
        except KeyError:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!") from None

    def __contains__(self, name):
        return name in self._table

    def __iter__(self):
        return iter(self._table.items())

    def keys(self):
        return self._table.keys()


DATASET_REGISTRY = Registry("dataset")
ARCH_REGISTRY = Registry("arch")
MODEL_REGISTRY = Registry("model")
LOSS_REGISTRY = Registry("loss")
METRIC_REGISTRY = Registry("metric")

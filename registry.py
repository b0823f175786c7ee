"""Name -> callable registries behind ``builder.export_from_registry``.

Drop-in for the reference's ``registry.py`` (reference registry.py:1-61): three
module-level registries (``cfg``, ``model``, ``trainer``) whose stored keys are
``"<registry name>_<key>"``.  Behaviour kept from the reference:

* ``reg("key")`` returns a decorator, a bare ``@reg`` registers under the
  callable's ``__name__`` (reference registry.py:41-56);
* a non-callable value raises ``Exception`` (reference registry.py:7-8);
* re-registering a key prints a warning and overwrites (reference registry.py:15-17);
* dict-style ``keys/values/items/in/[]`` and a read-only ``name``.
"""


class Register:
    def __init__(self, registry_name) -> None:
        self._name = registry_name
        self._dict = {}

    # -- storage -----------------------------------------------------------
    def _full_key(self, key, value):
        suffix = value.__name__ if key is None else key
        return f"{self._name}_{suffix}"

    def __setitem__(self, key, value):
        if not callable(value):
            raise Exception(f"注册的值：{value}必须是callable")
        full = self._full_key(key, value)
        if full in self._dict:
            print(f"警告：{full}已经存在，将被覆盖")
        self._dict[full] = value

    def __getitem__(self, key):
        return self._dict[key]

    def __contains__(self, key):
        return key in self._dict

    def __str__(self) -> str:
        return str(self._dict)

    def keys(self):
        return self._dict.keys()

    def values(self):
        return self._dict.values()

    def items(self):
        return self._dict.items()

    @property
    def name(self):
        return self._name

    # -- decorator entry points ---------------------------------------------
    def register(self, target):
        if callable(target):          # bare @registry
            self[None] = target
            return target

        def decorator(obj):           # @registry("key")
            self[target] = obj
            return obj

        return decorator

    __call__ = register


config_registry = Register("cfg")
model_registry = Register("model")
trainer_registry = Register("trainer")

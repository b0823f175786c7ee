"""``export_from_registry`` -- the plugin lookup train.py / predict.py call.

Same contract as the reference (builder.py:8-31): lower-case the name, validate
it against ``check.MODELS`` (``ValueError``), look up ``cfg_<n>``, ``model_<n>``,
``trainer_<n>`` (``KeyError`` if absent) and return
``(cfg instance, algorithm class, trainer class)``.  Registration happens by the
import side effects below, as in the reference (builder.py:2-5).  Models whose
MI355X path has not been built yet are whitelisted but unregistered, so they
fail with the reference's own ``KeyError``.
"""
from check import check_model_name
from registry import config_registry, model_registry, trainer_registry
import configs  # noqa: F401  (registers cfg_*)
import core.trainer  # noqa: F401  (registers trainer_*)
import core.algorithms  # noqa: F401  (registers model_*)


def export_from_registry(name: str):
    name = name.lower()
    check_model_name(name)
    found = []
    for registry, prefix in ((config_registry, "cfg_"), (model_registry, "model_"), (trainer_registry, "trainer_")):
        key = prefix + name
        if key not in registry:
            raise KeyError(f"找不到{registry.name}注册器中的key：{key}")
        found.append(registry[key])
    cfg_cls, algorithm_cls, trainer_cls = found
    return cfg_cls(), algorithm_cls, trainer_cls

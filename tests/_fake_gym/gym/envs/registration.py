"""TEST STAND-IN: ``register`` records (id -> entry point, kwargs) and refuses a second registration of an id like gym does;
``make`` resolves ``'module:attr'`` with importlib and calls it."""
import importlib

registry = {}
calls = []


class EnvSpec(object):
    def __init__(self, id, entry_point=None, kwargs=None, **other):
        self.id, self.entry_point, self._kwargs, self.other = id, entry_point, dict(kwargs or {}), other

    def make(self, **kwargs):
        mod_name, attr = self.entry_point.split(":")
        cls = getattr(importlib.import_module(mod_name), attr)
        kw = dict(self._kwargs)
        kw.update(kwargs)
        env = cls(**kw)
        env.spec = self
        return env


def register(id, **kwargs):
    calls.append((id, dict(kwargs)))
    if id in registry:
        raise RuntimeError("Cannot re-register id: %s" % id)
    registry[id] = EnvSpec(id, **kwargs)


def make(id, **kwargs):
    if id not in registry:
        raise KeyError("No registered env with id: %s" % id)
    return registry[id].make(**kwargs)

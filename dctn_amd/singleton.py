"""Metaclass that makes a class a process-wide singleton (reference: dctn/singleton.py:1-7)."""


class Singleton(type):
    _instances: dict = {}

    def __call__(cls, *args, **kwargs):
        inst = Singleton._instances.get(cls)
        if inst is None:
            inst = Singleton._instances[cls] = super().__call__(*args, **kwargs)
        return inst

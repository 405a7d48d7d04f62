"""Dataset registry with the reference's surface (test_phase/datasets/datasets.py:4-19)."""
import os

DEFAULT_ROOT = './materials'
datasets = {}


def register(name):
    def decorator(cls):
        datasets[name] = cls
        return cls
    return decorator


def make(name, **kwargs):
    if kwargs.get('root_path') is None:
        kwargs['root_path'] = os.path.join(DEFAULT_ROOT, name)
    return datasets[name](**kwargs)

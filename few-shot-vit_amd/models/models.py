"""Model registry with the reference's surface (test_phase/models/models.py:4-26):
`@register(name)`, `make(name, **kwargs)`, `load(ckpt_dict, name='model')`."""
import torch

models = {}


def register(name):
    def decorator(cls):
        models[name] = cls
        return cls
    return decorator


def make(name, **kwargs):
    """`make(None)` is None; unknown names raise KeyError; the model is moved to the GPU when one
    is visible (models.py:12-18)."""
    if name is None:
        return None
    model = models[name](**kwargs)
    if torch.cuda.is_available():
        model.cuda()
    return model


def load(model_sv, name=None):
    """Rebuild from a checkpoint dict {name, name_args, name_sd} (models.py:21-26; schema
    meta_tuning_sun_m/train_meta.py:241-257)."""
    if name is None:
        name = 'model'
    model = make(model_sv[name], **model_sv[name + '_args'])
    model.load_state_dict(model_sv[name + '_sd'])
    return model

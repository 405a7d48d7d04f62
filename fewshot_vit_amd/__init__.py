"""Importable alias of the `few-shot-vit_amd/` package directory.

The product package lives in `few-shot-vit_amd/` (the name the build contract fixes); a hyphen
is not a legal Python identifier, so this shim extends its own search path with that directory:
`import fewshot_vit_amd.models` resolves to `few-shot-vit_amd/models/`.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'few-shot-vit_amd')
__path__.insert(0, _real)
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))

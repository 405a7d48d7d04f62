"""few-shot-vit_amd: MI355X-native episodic few-shot ViT engine (hot path only).

Host side (Python, mirrors the reference's `models` / `utils` / `datasets.samplers` surface)
over a C-ABI shared library of hand-written gfx950 HIP kernels (`csrc/`, `include/fsvit.h`).
Import as `fewshot_vit_amd` (see the alias shim next to this directory).
"""
__version__ = '0.1.0'

#!/usr/bin/env python3
"""Meta-tuning with warm-up (meta_tuning_sun_m/train_meta_warmup.py): identical to `train_meta` except for
  * 500 instead of 200 `tval` / `val` batches per epoch (:95,:112),
  * `SGD(lr, momentum=0.9, weight_decay)` + `MultiStepLRScheduler(milestones, decay_rate=0.5, warmup_lr_init=1e-5, warmup_t=3)`
    (:140-141) stepped with `step(epoch - 1)` at the end of every epoch (:217) - so epoch 1 AND epoch 2 train at 1e-5, epoch
    e >= 2 at `_get_lr(e - 2)`: the reference's off-by-one is kept.
The scheduler is `utils.schedulers.MultiStepLRScheduler` (timm's algorithm restated; timm is not installed here).

  python -m fewshot_vit_amd.train_meta_warmup --config few-shot-vit_amd/configs/train_meta_synthetic.yaml
"""
from .train_meta import cli, main as _main


def main(config, *args, **kwargs):
    kwargs['warmup'] = True
    return _main(config, *args, **kwargs)


if __name__ == '__main__':
    cli(warmup=True)

from nasrec_amd.utils.lr_schedule import ConstantWithWarmup, CosineAnnealingWarmupRestarts  # noqa: F401

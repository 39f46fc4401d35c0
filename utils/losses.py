from transformertts_amd.utils.losses import masked_mean_absolute_error, weighted_sum_losses  # noqa: F401

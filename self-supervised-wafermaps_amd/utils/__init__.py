from . import benchmarking, debug, model_utils, scheduler  # noqa: F401
from .model_utils import (activate_requires_grad, batch_shuffle, batch_unshuffle, deactivate_requires_grad, get_at_index, mask_at_index,  # noqa: F401
                          patchify, random_token_mask, repeat_token, set_at_index, update_momentum)

"""rsrgan_amd -- MI355X-native (gfx950) implementation of RSRGAN's sequence-level GAN training step.

Drop-in for the hot path of wangkenpu/rsrgan: `GAN_RNN` (models/gan_rnn_placeholder.py) and
`train_one_iteration` (scripts/train_gan_rnn_placeholder.py:48-133).  All arithmetic runs in
hand-written HIP kernels behind the C ABI of include/rsrgan.h (rsrgan_amd/lib/librsrgan_hip.so);
there is NO CPU fallback: constructing a model without the library or without a GPU raises.
"""
from .gan_rnn import GAN_RNN, Model                      # noqa: F401
from .gan import GAN                                     # noqa: F401
from .trainer import RNNTrainer, DNNTrainer              # noqa: F401
from .segan import SEGAN                                 # noqa: F401
from .train import (train_one_iteration, eval_one_iteration,   # noqa: F401
                    exponential_decay)

__all__ = ["GAN_RNN", "GAN", "SEGAN", "RNNTrainer", "DNNTrainer", "Model", "train_one_iteration", "eval_one_iteration", "exponential_decay"]

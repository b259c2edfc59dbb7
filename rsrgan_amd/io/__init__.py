"""NumPy / Kaldi-ark data path replacing the reference's io_funcs (TFRecords + tf.data):
ark/scp reader and writer, global CMVN, Kaldi-style splicing, length-bucketed padded batches."""
from .kaldi_ark import ArkReader, ArkWriter, read_binary_file, convert_cmvn_to_numpy     # noqa: F401
from .features import apply_cmvn, splice_feats, PaddedBatchReader, FrameBatchReader                          # noqa: F401
from .prefetch import prefetch                                                             # noqa: F401

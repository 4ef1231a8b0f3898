"""Data side of the MI355X build: tokenizer contract, TFRecord / tf.train.Example codec, paired-dataset writer."""
from . import tfrecord  # noqa: F401
from .tokenizer_utils import get_tokenizer

__all__ = ["get_tokenizer", "tfrecord"]

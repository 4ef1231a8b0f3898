from .tokenizer_utils import get_tokenizer  # noqa: F401

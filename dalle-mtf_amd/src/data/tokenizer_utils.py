"""get_tokenizer (reference src/data/tokenizer_utils.py:4-16): GPT-2 fast tokenizer + '<|padding|>'.
The GPT-2 vocabulary is not available offline here; when it cannot be loaded a size-only stand-in keeps
the train_dalle.py contract (len(tokenizer) == text_vocab_size, padding id = last id)."""


class _OfflineTokenizer:
    def __init__(self, vocab_size=50258):
        self._n = vocab_size
        self.pad_token = "<|padding|>"

    def __len__(self):
        return self._n

    def encode(self, text):
        if text == self.pad_token:
            return [self._n - 1]
        raise RuntimeError("offline stand-in tokenizer: only the padding token can be encoded")


def get_tokenizer(tokenizer_type=None, from_pretrained=True, add_padding_token=True, vocab_size=50258):
    if tokenizer_type is None or (tokenizer_type.lower() == "hf_gpt2tokenizerfast" and from_pretrained):
        try:
            from transformers import GPT2TokenizerFast
            tok = GPT2TokenizerFast.from_pretrained("gpt2", local_files_only=True)
            if add_padding_token:
                tok.add_special_tokens({'pad_token': '<|padding|>'})
            if len(tok) > 2:
                return tok
        except Exception:
            pass
        return _OfflineTokenizer(vocab_size)
    raise NotImplementedError(f"{tokenizer_type} / {from_pretrained} not implemented")

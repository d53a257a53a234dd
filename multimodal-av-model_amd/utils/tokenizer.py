"""Character tokenizer over a sentencepiece ``.vocab`` file (utils/tokenizer.py:3-41): id = line index,
per-CHARACTER lookup with ' ' -> '▁'; ``<blank>`` is line 3 of tokenizer800.vocab, ``<unk>`` line 0."""
from __future__ import annotations


class Tokenizer:
    def __init__(self, vocab_path):
        self.id_to_token = []
        self.token_to_id = {}
        with open(vocab_path, "r", encoding="utf-8") as f:
            for idx, line in enumerate(f):
                tok = line.strip().split("\t")[0]
                self.token_to_id[tok] = idx          # later duplicates win, as in the reference
                self.id_to_token.append(tok)

    def encode(self, text):
        unk = self.unk_id
        return [self.token_to_id.get("▁" if ch == " " else ch, unk) for ch in text]

    def decode(self, ids):
        n = len(self.id_to_token)
        return "".join(self.id_to_token[i] for i in ids if 0 <= i < n).replace("▁", " ").strip()

    @property
    def vocab_size(self):
        return len(self.id_to_token)

    @property
    def pad_id(self):
        return self.token_to_id.get("<pad>", 0)

    @property
    def blank_id(self):
        return self.token_to_id.get("<blank>", 0)

    @property
    def unk_id(self):
        return self.token_to_id.get("<unk>", 0)


class SyntheticTokenizer(Tokenizer):
    """800-entry stand-in with the same special ids (<unk>0 <s>1 </s>2 <blank>3 ▁4) for synthetic benchmarks."""

    def __init__(self, vocab_size: int = 800):
        self.id_to_token = ["<unk>", "<s>", "</s>", "<blank>", "▁"] + [chr(0xAC00 + i) for i in range(vocab_size - 5)]
        self.token_to_id = {t: i for i, t in enumerate(self.id_to_token)}

"""Character-level tokenizer over a sentencepiece ``.vocab`` file — behaviour of the reference's
``utils/tokenizer.py:3-41``: the id of a piece is its line index; text is encoded per CHARACTER (a space is looked up
as '▁'), unknown characters map to ``<unk>``; in tokenizer800.vocab ``<unk>``=0, ``<s>``=1, ``</s>``=2, ``<blank>``=3, '▁'=4.
"""
from __future__ import annotations

from typing import Iterable, List

_SPACE = "▁"


def _read_pieces(vocab_path: str) -> List[str]:
    pieces = []
    with open(vocab_path, "r", encoding="utf-8") as fh:
        for raw in fh:
            pieces.append(raw.strip().split("\t")[0])          # "<piece>\t<score>"
    return pieces


class Tokenizer:
    _SPECIALS = {"pad_id": "<pad>", "blank_id": "<blank>", "unk_id": "<unk>"}

    def __init__(self, vocab_path):
        self._install(_read_pieces(vocab_path))

    def _install(self, pieces: List[str]) -> None:
        self.id_to_token = list(pieces)
        self.token_to_id = {}
        for i, tok in enumerate(self.id_to_token):              # a repeated piece keeps its LAST index
            self.token_to_id[tok] = i

    def __getattr__(self, name):                                 # pad_id / blank_id / unk_id: index of the special piece, else 0
        specials = type(self)._SPECIALS
        if name in specials and "token_to_id" in self.__dict__:
            return self.token_to_id.get(specials[name], 0)
        raise AttributeError(name)

    @property
    def vocab_size(self) -> int:
        return len(self.id_to_token)

    def encode(self, text: str) -> List[int]:
        lookup, unk = self.token_to_id, self.unk_id
        return [lookup.get(_SPACE if ch == " " else ch, unk) for ch in text]

    def decode(self, ids: Iterable[int]) -> str:
        table = self.id_to_token
        return "".join(table[i] for i in ids if 0 <= i < len(table)).replace(_SPACE, " ").strip()


class SyntheticTokenizer(Tokenizer):
    """800-entry stand-in with the same special ids for synthetic benchmarks (no vocab file needed)."""

    def __init__(self, vocab_size: int = 800):
        self._install(["<unk>", "<s>", "</s>", "<blank>", _SPACE] + [chr(0xAC00 + i) for i in range(vocab_size - 5)])

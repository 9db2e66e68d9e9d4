"""CLIP byte-pair tokenizer for the text-prompt path (host side, T0 of SURVEY.md §8a).

Behaviour follows the reference's ``SimpleTokenizer`` (sam3/sam3/model/tokenizer_ve.py:128-253):
"lower" cleaning (ftfy -> double html.unescape -> whitespace collapse -> lower case), the CLIP
pre-tokenisation regex, byte-level BPE with the merge list of ``bpe_simple_vocab_16e6.txt.gz``
(lines 1 .. 48894 of the file), ``<start_of_text>`` / ``<end_of_text>`` = 49406 / 49407,
zero padding to ``context_length`` and truncation with the last token forced to EOT.

The merge table is a data asset of the reference (``sam3/assets/``); it is not shipped here --
pass ``bpe_path`` (the reference's builder takes the same argument, model_builder.py:944-946).
"""
from __future__ import annotations

import gzip
import html
import os
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import regex

try:  # ftfy only matters for mojibake / exotic unicode; ASCII prompts pass through unchanged
    import ftfy
    _fix_text = ftfy.fix_text
except Exception:  # pragma: no cover - not installed in the build image
    def _fix_text(t: str) -> str:
        return t

SOT, EOT = "<start_of_text>", "<end_of_text>"
N_MERGES = 49152 - 256 - 2
_PATTERN = regex.compile(
    r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)


def _byte_symbols() -> Dict[int, str]:
    """Printable stand-in character for every byte value (the GPT-2 / CLIP table): bytes that are
    already printable map to themselves, the remaining ones to code points 256, 257, ..."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class ClipBpeTokenizer:
    def __init__(self, bpe_path: Union[str, os.PathLike], context_length: int = 77):
        if bpe_path is None or not os.path.exists(bpe_path):
            raise FileNotFoundError(
                f"BPE merge table not found: {bpe_path!r} (pass the reference's assets/bpe_simple_vocab_16e6.txt.gz)")
        with gzip.open(bpe_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(ln.split()) for ln in lines[1: N_MERGES + 1]]
        self.byte_symbol = _byte_symbols()
        # the vocabulary is ordered by the byte table's construction order (printable first)
        order = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
        order += [b for b in range(256) if b not in order]
        base = [self.byte_symbol[b] for b in order]
        vocab = base + [c + "</w>" for c in base] + ["".join(m) for m in merges] + [SOT, EOT]
        self.token_id = {t: i for i, t in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.sot_id, self.eot_id = self.token_id[SOT], self.token_id[EOT]
        self.context_length = context_length
        self._memo: Dict[str, Tuple[str, ...]] = {}

    @staticmethod
    def clean(text: str) -> str:
        text = html.unescape(html.unescape(_fix_text(text))).strip()
        return regex.sub(r"\s+", " ", text).strip().lower()

    def _merge_word(self, word: str) -> Tuple[str, ...]:
        """Greedy lowest-rank-first pair merging of one pre-token (symbols of the byte table)."""
        hit = self._memo.get(word)
        if hit is not None:
            return hit
        parts: List[str] = list(word[:-1]) + [word[-1] + "</w>"]
        while len(parts) > 1:
            best, best_rank = None, None
            for a, b in zip(parts, parts[1:]):
                r = self.rank.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            merged, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                    merged.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        out = tuple(parts)
        self._memo[word] = out
        return out

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for tok in _PATTERN.findall(self.clean(text)):
            if tok in (SOT, EOT):
                ids.append(self.token_id[tok])
                continue
            word = "".join(self.byte_symbol[b] for b in tok.encode("utf-8"))
            ids.extend(self.token_id[p] for p in self._merge_word(word))
        return ids

    def __call__(self, texts: Union[str, Sequence[str]], context_length: Optional[int] = None) -> np.ndarray:
        """-> int64 [n_texts, context_length], zero padded; over-long texts are truncated and end in EOT."""
        if isinstance(texts, str):
            texts = [texts]
        n = context_length or self.context_length
        assert n, "Please set a valid context length"
        out = np.zeros((len(texts), n), dtype=np.int64)
        for i, t in enumerate(texts):
            ids = [self.sot_id] + self.encode(t) + [self.eot_id]
            if len(ids) > n:
                ids = ids[:n]
                ids[-1] = self.eot_id
            out[i, : len(ids)] = ids
        return out

"""CLIP byte-pair tokenizer for the expressions (reference utils/simple_tokenizer.py:62-133 `SimpleTokenizer`,
utils/dataset.py:43-84 `tokenize`).  Index work: the contract is BIT-EXACT token ids.

Host code (the reference's is Python too).  The tokenizer is defined by data the reference ships and this repository does
not copy: the merge list `bpe_simple_vocab_16e6.txt.gz` (OpenAI CLIP).  Pass its path (`BPETokenizer(path)`), or set
CRIS_BPE_PATH; `default_merges_path()` also looks into a reference checkout next to the working directory.

Definition (what any implementation has to reproduce):
  * vocabulary: 256 byte symbols in the order "printable Latin-1 first" (33..126, 161..172, 174..255, then the other 68 byte
    values in increasing order, which are written as the code points 256..323), the same 256 with the end-of-word mark `</w>`,
    the first 48 894 merges of the file in file order, `<|startoftext|>` = 49406, `<|endoftext|>` = 49407;
  * text: ftfy.fix_text, html.unescape twice, strip, runs of white space -> one blank, lower case;
  * words: the regular expression below (contractions, letter runs, single digits, runs of other non-space characters);
  * a word's UTF-8 bytes become symbols, the last one carries `</w>`; then, while some adjacent pair is in the merge list, every
    occurrence of the pair with the LOWEST rank is merged, left to right.
`tokenize` pads with 0 to the context length and, when asked to truncate, overwrites the last kept id with `<|endoftext|>`
(so that `word.argmax(-1)` still finds the end-of-text position, model/clip.py:451-452).

ftfy is not installed everywhere.  Its repair is the identity on printable ASCII, which is what the RefCOCO / RefCOCO+ / G-Ref
expressions are; with ftfy absent a non-ASCII text raises instead of being tokenized differently from the reference.
"""
import gzip
import html
import os
from typing import Callable, Dict, List, Optional, Sequence, Union

import torch

SOT, EOT = "<|startoftext|>", "<|endoftext|>"
N_MERGES = 49152 - 256 - 2
WORD_PATTERN = r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"


def default_merges_path() -> Optional[str]:
    for p in (os.environ.get("CRIS_BPE_PATH"), os.path.join("utils", "bpe_simple_vocab_16e6.txt.gz"),
              "/root/reference/utils/bpe_simple_vocab_16e6.txt.gz"):
        if p and os.path.isfile(p):
            return p
    return None


def _byte_symbols() -> List[str]:
    """symbol of every byte value: printable Latin-1 bytes stand for themselves, the other 68 get the code points 256.."""
    plain = set(range(33, 127)) | set(range(161, 173)) | set(range(174, 256))
    table, extra = [], 0
    for b in range(256):
        if b in plain:
            table.append(chr(b))
        else:
            table.append(chr(256 + extra))
            extra += 1
    return table


def _ascii_only_repair(text: str) -> str:
    if all(32 <= ord(ch) < 127 or ch in "\t\n\r" for ch in text):
        return text
    raise RuntimeError("this text has non-ASCII characters and the reference repairs those with ftfy.fix_text, which is not "
                       "installed: install ftfy (or pass fix_text=) - refusing to tokenize differently from the reference: %r" % text[:60])


class BPETokenizer:
    def __init__(self, merges_path: Optional[str] = None, fix_text: Optional[Callable[[str], str]] = None):
        import regex
        merges_path = merges_path or default_merges_path()
        if merges_path is None:
            raise FileNotFoundError("CLIP merge list not found: pass the path of bpe_simple_vocab_16e6.txt.gz (the reference ships "
                                    "it under utils/) or set CRIS_BPE_PATH")
        opener = gzip.open if merges_path.endswith(".gz") else open
        with opener(merges_path, "rb") as f:
            lines = f.read().decode("utf-8").split("\n")
        pairs = [tuple(ln.split()) for ln in lines[1:1 + N_MERGES]]            # line 0 is a version header
        if len(pairs) != N_MERGES or any(len(p) != 2 for p in pairs):
            raise ValueError("%s does not look like the CLIP merge list (%d usable lines)" % (merges_path, len(pairs)))
        sym = _byte_symbols()
        # vocabulary order of the byte symbols: the printable ones in byte order, then the remapped ones in byte order
        order = [b for b in range(256) if ord(sym[b]) < 256] + [b for b in range(256) if ord(sym[b]) >= 256]
        vocab = [sym[b] for b in order]
        vocab = vocab + [v + "</w>" for v in vocab] + [a + b for a, b in pairs] + [SOT, EOT]
        self.ids: Dict[str, int] = {v: i for i, v in enumerate(vocab)}
        assert len(self.ids) == len(vocab) == 49408
        self.words = vocab
        self.rank: Dict[tuple, int] = {p: i for i, p in enumerate(pairs)}
        self.byte_sym = sym
        self.sym_byte = {s: b for b, s in enumerate(sym)}
        self.sot, self.eot = self.ids[SOT], self.ids[EOT]
        self._pat = regex.compile(WORD_PATTERN, regex.IGNORECASE)
        self._space = regex.compile(r"\s+")
        self._cache: Dict[str, List[int]] = {}
        if fix_text is None:
            try:
                import ftfy
                fix_text = ftfy.fix_text
            except ImportError:
                fix_text = _ascii_only_repair
        self._fix = fix_text

    # ---- one word -----------------------------------------------------------------------------------------------------
    def _merge(self, symbols: List[str]) -> List[str]:
        rank = self.rank
        while len(symbols) > 1:
            best, best_rank = None, None
            for a, b in zip(symbols, symbols[1:]):
                r = rank.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            a, b = best
            out, i, n = [], 0, len(symbols)
            while i < n:
                if i + 1 < n and symbols[i] == a and symbols[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(symbols[i])
                    i += 1
            symbols = out
        return symbols

    def _word_ids(self, word: str) -> List[int]:
        hit = self._cache.get(word)
        if hit is not None:
            return hit
        if word in (SOT, EOT):
            out = [self.ids[word]]
        else:
            symbols = [self.byte_sym[b] for b in word.encode("utf-8")]
            symbols[-1] += "</w>"
            out = [self.ids[s] for s in self._merge(symbols)]
        self._cache[word] = out
        return out

    # ---- text -----------------------------------------------------------------------------------------------------------
    def clean(self, text: str) -> str:
        text = html.unescape(html.unescape(self._fix(text))).strip()
        return self._space.sub(" ", text).strip().lower()

    def encode(self, text: str) -> List[int]:
        out: List[int] = []
        for word in self._pat.findall(self.clean(text)):
            out.extend(self._word_ids(word))
        return out

    def decode(self, ids: Sequence[int]) -> str:
        text = "".join(self.words[int(i)] for i in ids)
        return bytearray(self.sym_byte[ch] for ch in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def tokenize(self, texts: Union[str, Sequence[str]], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
        """int64 [len(texts), context_length]: <|startoftext|> ids <|endoftext|> 0 0 ... (utils/dataset.py:43-84)"""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, text in enumerate(texts):
            ids = [self.sot] + self.encode(text) + [self.eot]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError("Input %s is too long for context length %d" % (text, context_length))
                ids = ids[:context_length]
                ids[-1] = self.eot
            out[i, :len(ids)] = torch.tensor(ids, dtype=torch.long)
        return out

"""BPE tokenizer (cris/pytorch_amd/tokenizer.py) against the reference's own (utils/simple_tokenizer.py + utils/dataset.py
tokenize), bit exact: imported from /root/reference where that exists (this container), and against the committed known
answers (tests/golden/tokenizer_vectors.json, written by the reference tokenizer) wherever the CLIP merge list can be found.
The merge list is the reference's data file and is not part of this repository: without it these tests skip."""
import json
import os
import random

import pytest
import torch

from conftest import GOLDEN
from cris.pytorch_amd import tokenizer as T

MERGES = T.default_merges_path()
pytestmark = pytest.mark.skipif(MERGES is None, reason="CLIP merge list (bpe_simple_vocab_16e6.txt.gz) not available")


@pytest.fixture(scope="module")
def tok():
    return T.BPETokenizer(MERGES, fix_text=lambda s: s)          # (the reference is imported with the same ftfy stand-in)


@pytest.fixture(scope="module")
def ref_tokenize():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN)))
    from golden import ref_harness
    if not ref_harness.reference_available():
        pytest.skip("reference checkout not present")
    ref_harness.import_reference()
    from utils import dataset
    return dataset


def test_known_answers(tok):
    for v in json.load(open(os.path.join(GOLDEN, "tokenizer_vectors.json"))):
        ids = tok.tokenize(v["text"], v["context_length"], True)[0]
        assert ids.tolist() == v["ids"], v["text"]
        assert int(ids.argmax()) == v["argmax"]
    assert (tok.sot, tok.eot, len(tok.words)) == (49406, 49407, 49408)


WORDS = ("the man in red shirt on left woman's umbrella giraffe's head 2nd from right guy w/ hat it's they're don't i'll we've i'm she'd "
         "skateboarder blue-ish jacket #3 person (partially hidden) behind bench... zebra closest 2 us pizza w/o pepperoni ??? ok "
         "Bus 42 A&W logo 100% café naïve über señor 東京 ÅNGSTRÖM ﬁne x² ½ cup").split()


def _sentences(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        k = rng.randint(1, 24)
        parts = [rng.choice(WORDS) for _ in range(k)]
        sep = rng.choice([" ", " ", "  ", " \t", "\n"])
        s = sep.join(parts)
        if rng.random() < 0.2:
            s = "  " + s.upper() + " "
        if rng.random() < 0.15:
            s = s.replace("&", "&amp;").replace("w/", "w&#47;")
        if rng.random() < 0.1:
            s += " " + "".join(rng.choice("abcdefghijklmnopqrstuvwxyz0123456789'-") for _ in range(rng.randint(1, 30)))
        out.append(s)
    return out


def test_equals_reference_tokenizer(tok, ref_tokenize):
    texts = _sentences(3000, 0) + ["", " ", "a", "<|startoftext|> hi <|endoftext|>", "it's", "'s", "''''", "x" * 200, "1234567890",
                                    "&amp;amp; &lt;b&gt;", "tab\there", "émigré's café"]
    for L in (17, 22, 77):
        ref = ref_tokenize.tokenize(texts, L, True)
        got = tok.tokenize(texts, L, True)
        bad = (ref != got).any(dim=1).nonzero().flatten().tolist()
        assert not bad, [(texts[i], ref[i].tolist(), got[i].tolist()) for i in bad[:3]]
    # no truncation: same error behaviour
    with pytest.raises(RuntimeError, match="too long"):
        tok.tokenize("x " * 40, 17, False)
    with pytest.raises(RuntimeError, match="too long"):
        ref_tokenize.tokenize("x " * 40, 17, False)
    # the raw id lists and the round trip
    ref_tok = ref_tokenize._tokenizer
    for t in texts[:300]:
        ids = tok.encode(t)
        assert ids == ref_tok.encode(t)
        assert tok.decode(ids) == ref_tok.decode(ids)


def test_without_ftfy_non_ascii_is_refused_not_mangled():
    t = T.BPETokenizer(MERGES)
    try:
        import ftfy  # noqa: F401
        has = True
    except ImportError:
        has = False
    if has and getattr(__import__("ftfy"), "__file__", None):
        pytest.skip("ftfy installed")
    if t._fix is T._ascii_only_repair:
        assert t.tokenize("a plain ascii sentence", 17, True).shape == (1, 17)
        with pytest.raises(RuntimeError, match="ftfy"):
            t.tokenize("café", 17, True)


def test_property_any_text_tokenizes_like_the_reference(tok, ref_tokenize):
    """hypothesis: arbitrary unicode text (control characters, surrogates excluded), any context length"""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=400, deadline=None, derandomize=True)
    @given(st.text(alphabet=st.characters(blacklist_categories=("Cs",)), max_size=60), st.sampled_from([8, 17, 22, 77]))
    def check(text, L):
        ref = ref_tokenize.tokenize(text, L, True)
        got = tok.tokenize(text, L, True)
        assert torch.equal(ref, got), (text, ref.tolist(), got.tolist())

    check()

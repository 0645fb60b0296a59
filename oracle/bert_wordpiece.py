"""ORACLE — TEST INFRASTRUCTURE ONLY. Pure-Python restatement of BERT's uncased tokenizer (BasicTokenizer + WordpieceTokenizer).

The reference tokenises captions with `transformers.BertTokenizer.from_pretrained("bert-base-uncased")(caption, padding=False,
truncation=True, max_length=L)` (reference data/dataloader.py:139-141, 196-202; transformers==4.6.1 pinned, requirements.txt:99 — its slow
BertTokenizer is the published algorithm of google-research/bert `tokenization.py`, restated here from that publication):

  BasicTokenizer   1. clean: drop U+0000, U+FFFD and control characters (category C*, except \\t \\n \\r which count as whitespace);
                      every whitespace character (\\t \\n \\r, category Zs) becomes one space
                   2. put spaces around every CJK ideograph (the code-point ranges of the publication)
                   3. split on whitespace; per token: lower-case, NFD-decompose and drop category-Mn marks (do_lower_case=True)
                   4. split every token at punctuation: each punctuation character (ASCII 33-47, 58-64, 91-96, 123-126, or Unicode
                      category P*) is a token of its own
  WordpieceTokenizer  greedy longest-match-first from the left; non-initial pieces carry "##"; a word with no match at some position, or
                      longer than 100 characters, is ONE [UNK]
  framing          [CLS] tokens [SEP]; truncation=True with max_length=L keeps the first L - 2 tokens (single sequence)

Pinning: `tests/golden/make_tokens_golden.py` requires this restatement and the installed `transformers.BertTokenizer` (5.x, built on the
`tokenizers` library) to produce identical ids on every golden case before it writes the fixture; the product
(`clip_lite_amd.data.WordPieceTokenizer`) is then held to the fixture bit-exactly. Only tests/ may import this module."""
import unicodedata


def _is_whitespace(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F
            or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


def basic_tokenize(text):
    out = []
    for ch in text:
        cp = ord(ch)
        if cp == 0 or cp == 0xFFFD or _is_control(ch):
            continue
        out.append(" " if _is_whitespace(ch) else ch)
    text = "".join(out)
    out = []
    for ch in text:
        out.append(f" {ch} " if _is_cjk(ord(ch)) else ch)
    tokens = []
    for tok in "".join(out).split():
        tok = tok.lower()
        tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
        cur = ""
        for ch in tok:
            if _is_punctuation(ch):
                if cur:
                    tokens.append(cur)
                    cur = ""
                tokens.append(ch)
            else:
                cur += ch
        if cur:
            tokens.append(cur)
    return " ".join(tokens).split()


def wordpiece(word, vocab, unk="[UNK]", max_chars=100):
    if len(word) > max_chars:
        return [unk]
    pieces, start = [], 0
    while start < len(word):
        end, cur = len(word), None
        while start < end:
            sub = ("##" if start > 0 else "") + word[start:end]
            if sub in vocab:
                cur = sub
                break
            end -= 1
        if cur is None:
            return [unk]
        pieces.append(cur)
        start = end
    return pieces


def load_vocab(path):
    vocab = {}
    with open(path, encoding="utf-8") as fh:
        for i, line in enumerate(fh):
            tok = line.rstrip("\n")
            if tok != "" and tok not in vocab:
                vocab[tok] = i
    return vocab


def encode(text, vocab, max_length):
    """input_ids of `BertTokenizer(text, padding=False, truncation=True, max_length=max_length)`."""
    toks = [p for w in basic_tokenize(text) for p in wordpiece(w, vocab)]
    toks = toks[:max(max_length - 2, 0)]
    return [vocab["[CLS]"]] + [vocab[t] for t in toks] + [vocab["[SEP]"]]

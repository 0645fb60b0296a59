"""Token-id golden vectors for SURVEY §8f N3 (the caption side of the input pipeline), made in the BUILD container.

The reference tokenises with `transformers.BertTokenizer.from_pretrained("bert-base-uncased")(caption, padding=False, truncation=True,
max_length=L)` (reference data/dataloader.py:139-141, 196-202) after `NormalizeCaption` (data/transforms.py:46-90). The hub vocabulary is not
reachable here, so this script builds a WordPiece vocabulary from the reference's own caption fixture, `data/mock_data.json` (41 captions):
whole words for most, prefix + "##" continuation pieces for a deterministic subset, single letters / digits and their "##" forms as the tail,
a few letters left OUT so that some words end in [UNK], and punctuation entries. It then runs `transformers.BertTokenizer(vocab=...)` —
the class the reference's call resolves to (5.x here, where `vocab` replaced 4.x's `vocab_file`) — and, as a second independent checker, the
pure-Python restatement of the published BERT tokenizer (oracle/bert_wordpiece.py; the two must agree on every case) on every mock caption plus a set of adversarial strings (accents, punctuation, CJK, control
characters, over-long words, truncation at several lengths), and stores

    tests/golden/vocab_mock.txt     the vocabulary (generated data)
    tests/golden/tokens_mock.npz    captions (as given), max_length per case, the checker's input_ids (ragged, -1 padded)

`tests/test_host_logic.py::test_wordpiece_matches_transformers_bert_tokenizer_golden` requires `clip_lite_amd.data.WordPieceTokenizer` to
reproduce every id: integer work, no tolerance. Only numbers and the caption strings (data) are stored, no reference source.

    python tests/golden/make_tokens_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def build_vocab(captions):
    from clip_lite_amd.data import normalize_caption
    words = []
    for c in captions:
        for w in normalize_caption(c, 30).split():
            if w not in words:
                words.append(w)
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(3)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    missing = set("qxz")                                     # letters with no single-character entry: words that need them end in [UNK]
    for ch in "abcdefghijklmnopqrstuvwxyz0123456789":
        if ch not in missing:
            vocab += [ch, "##" + ch]
    vocab += ["!", ",", ".", "?", "'", "-", "(", ")", ":", ";", "&", "中", "文"]
    for w in words:
        h = int.from_bytes(hashlib.md5(w.encode()).digest()[:4], "little")
        if len(w) >= 6 and h % 3 == 0:                        # a third of the longer words only as prefix + continuation pieces
            cut = 2 + h % (len(w) - 3)
            for piece in (w[:cut], "##" + w[cut:]):
                if piece not in vocab:
                    vocab.append(piece)
        elif len(w) >= 5 and h % 3 == 1:                      # another third: a shorter prefix exists too (greedy must take the longest match)
            for piece in (w[:3], w, "##" + w[3:]):
                if piece not in vocab:
                    vocab.append(piece)
        elif h % 7 != 0:                                      # the rest whole, except a few left to the single letters
            if w not in vocab:
                vocab.append(w)
    vocab += ["##s", "##ing", "##ed", "cafe", "skate", "board", "un", "##believ", "##able"]
    seen, out = set(), []
    for t in vocab:
        if t not in seen:
            seen.add(t)
            out.append(t)
    return out


EXTRA = [
    ("A man, riding a skate-board!", 30), ("two dogs, zebra at Café!", 30), ("unbelievable quiz boxes", 30),
    ("it's (really) *great*: no?", 30), ("中文 mixed with english", 30), ("tab\there\x00and�control", 30),
    ("a" * 120 + " short", 30), ("", 30), ("   ", 30), ("naïve résumé façade", 30),
    ("the cat " * 25, 30), ("the cat " * 25, 8), ("the cat " * 25, 3), ("the cat " * 25, 2), ("photography" * 3, 30),
    ("Hello, World! HELLO world.", 16), ("3d globes & 2 hands - 1 globe", 30),
]


def main():
    from transformers import BertTokenizer
    from clip_lite_amd.data import normalize_caption
    with open("/root/reference/data/mock_data.json") as fh:
        captions = [r["caption"] if isinstance(r["caption"], str) else r["caption"][0] for r in json.load(fh)]
    vocab = build_vocab(captions)
    vp = os.path.join(HERE, "vocab_mock.txt")
    with open(vp, "w", encoding="utf-8") as fh:
        fh.write("\n".join(vocab) + "\n")
    from oracle import bert_wordpiece as W
    vdict = W.load_vocab(vp)
    # transformers 5.x: `vocab` (a dict) replaced 4.x's `vocab_file`; bert-base-uncased's settings (lower-case, accents stripped with it)
    tk = BertTokenizer(vocab=vdict, do_lower_case=True)
    assert tk.cls_token_id == vdict["[CLS]"] and tk.sep_token_id == vdict["[SEP]"] and tk.unk_token_id == vdict["[UNK]"]
    cases = [(c, 30) for c in captions] + [(c, 12) for c in captions[:10]] + EXTRA
    ids = []
    for cap, L in cases:
        # the reference's order (data/dataloader.py:194-202): NormalizeCaption first, then the tokenizer
        enc = tk(normalize_caption(cap, L), padding=False, truncation=True, max_length=L)
        ids.append(list(enc["input_ids"]))
    # and the tokenizer on RAW strings (no NormalizeCaption): pins the BertNormalizer / pre-tokenizer rules themselves
    raw = [list(tk(cap, padding=False, truncation=True, max_length=L)["input_ids"]) for cap, L in EXTRA]
    # second, independent checker: the pure-Python restatement of the published BasicTokenizer + WordpieceTokenizer (oracle/bert_wordpiece.py)
    for (cap, L), want in zip(cases, ids):
        got = W.encode(normalize_caption(cap, L), vdict, L)
        assert got == want, ("restatement vs transformers", cap, L, got, want)
    for (cap, L), want in zip(EXTRA, raw):
        got = W.encode(cap, vdict, L)
        assert got == want, ("restatement vs transformers (raw)", cap, L, got, want)
    width = max(len(r) for r in ids + raw)
    pad = lambda rows: np.array([r + [-1] * (width - len(r)) for r in rows], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "tokens_mock.npz"), captions=np.array([c for c, _ in cases]), max_length=np.array([L for _, L in cases], dtype=np.int32),
                        input_ids=pad(ids), raw_captions=np.array([c for c, _ in EXTRA]), raw_max_length=np.array([L for _, L in EXTRA], dtype=np.int32),
                        raw_input_ids=pad(raw))
    n_unk = sum(r.count(vocab.index("[UNK]")) for r in ids)
    n_cont = sum(1 for r in ids for t in r if t >= 0 and vocab[t].startswith("##"))
    print(f"{len(cases)} + {len(EXTRA)} cases, vocab {len(vocab)}, {n_unk} [UNK], {n_cont} continuation pieces, longest {width}")


if __name__ == "__main__":
    main()

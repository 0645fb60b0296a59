"""Batch sources for the pretraining step. Only the batch-dict CONTRACT of the reference's input pipeline is in scope
(SURVEY.md §8b): `image` f32 NCHW (ImageNet-normalised), `input_ids` / `attention_mask` int64 right-padded with the pad id 0
(reference data/dataloader.py:218-236), or `caption_encodings` f32 [B][768] in the frozen-sentence-embedding mode
(reference model.py:48-50). The LMDB / albumentations / tokenizer stack (reference data/*.py) is CPU-side I/O outside the
accelerated path and its dependencies are absent from the image.

`RandomDataset` is the counterpart of the reference's synthetic dataset (data/dataloader.py:36-114: random 3x224x224 images and
four fixed captions, len 118000); captions are mapped to token ids by a deterministic hash because the HF tokenizer files are
not available offline. `JsonCaptionDataset` reads the reference's json record format ({"image": path, "caption": str},
data/mock_data.json) for its captions; image files referenced there do not exist on any machine we have, so images are
synthetic.
"""
import hashlib
import json
import re
import unicodedata

import torch
from torch.utils.data import Dataset

CAPTIONS = ["a photo of a cat sitting on a couch", "two people riding bikes down a city street",
            "a plate of food with vegetables on a table", "a large airplane flying through a cloudy sky"]


_DROPPED = str.maketrans("", "", ",.'!?\"()*#:;~")


def normalize_caption(caption: str, max_caption_length: int = 30) -> str:
    """The reference's caption normalisation (data/transforms.py:46-90, NormalizeCaption): lower-case; delete , . ' ! ? " ( ) * # : ; ~;
    '-' and '/' become spaces; the COCO placeholder "<person>" becomes "person"; runs of two or more whitespace characters collapse
    to one space; trailing newlines and surrounding spaces go; at most `max_caption_length` space-separated words are kept; finally NFKD
    decomposition with the combining marks removed (accents stripped)."""
    c = caption.lower().translate(_DROPPED).replace("-", " ").replace("/", " ").replace("<person>", "person")
    c = re.sub(r"\s{2,}", " ", c).rstrip("\n").strip(" ")
    words = c.split(" ")
    if len(words) > max_caption_length:
        c = " ".join(words[:max_caption_length])
    c = unicodedata.normalize("NFKD", c.lower())
    return "".join(ch for ch in c if not unicodedata.combining(ch))


def hash_tokenize(caption: str, max_len: int, vocab: int = 30522):
    """normalize_caption, then map every word to a stable id in [1000, vocab); [CLS]=101 ... [SEP]=102 (the HF tokenizer files the
    reference loads, data/tokenizers.py, are not available offline)."""
    words = re.sub(r"[^a-z0-9 ]", " ", normalize_caption(caption, max_len)).split()
    ids = [101] + [1000 + int.from_bytes(hashlib.md5(w.encode()).digest()[:4], "little") % (vocab - 1000) for w in words][: max_len - 2] + [102]
    return ids


class _CaptionDataset(Dataset):
    def __init__(self, mode: str, image_size: int, max_caption_length: int, length: int, seed: int = 0):
        self.mode, self.image_size, self.max_len, self.length, self.seed = mode, image_size, max_caption_length, length, seed

    def __len__(self):
        return self.length

    def caption(self, idx):
        raise NotImplementedError

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        item = {"image_id": torch.tensor(idx, dtype=torch.long), "image": torch.randn(3, self.image_size, self.image_size, generator=g)}
        if self.mode == "sbert":
            item["caption_encodings"] = torch.randn(768, generator=g)
        else:
            item["caption_tokens"] = torch.tensor(hash_tokenize(self.caption(idx), self.max_len), dtype=torch.long)
        return item

    def collate_fn(self, items):
        batch = {"image_id": torch.stack([i["image_id"] for i in items]), "image": torch.stack([i["image"] for i in items])}
        if self.mode == "sbert":
            batch["caption_encodings"] = torch.stack([i["caption_encodings"] for i in items])
        else:
            L = max(len(i["caption_tokens"]) for i in items)
            ids = torch.zeros(len(items), L, dtype=torch.long)             # pad_token_id = 0
            mask = torch.zeros(len(items), L, dtype=torch.long)
            for r, i in enumerate(items):
                n = len(i["caption_tokens"])
                ids[r, :n] = i["caption_tokens"]
                mask[r, :n] = 1
            batch["input_ids"], batch["attention_mask"] = ids, mask
        return batch


class RandomDataset(_CaptionDataset):
    def __init__(self, mode="train_sbert", image_size=224, max_caption_length=30, length=118000, seed=0):
        super().__init__(mode, image_size, max_caption_length, length, seed)

    def caption(self, idx):
        return CAPTIONS[idx % len(CAPTIONS)]


class JsonCaptionDataset(_CaptionDataset):
    def __init__(self, json_files, mode="train_sbert", image_size=224, max_caption_length=30, seed=0):
        self.records = []
        for f in json_files:
            with open(f) as fh:
                self.records += json.load(fh)
        super().__init__(mode, image_size, max_caption_length, len(self.records), seed)

    def caption(self, idx):
        c = self.records[idx]["caption"]
        return c[0] if isinstance(c, list) else c

"""Batch sources for the pretraining step. Only the batch-dict CONTRACT of the reference's input pipeline is in scope
(SURVEY.md §8b): `image` f32 NCHW (ImageNet-normalised), `input_ids` / `attention_mask` int64 right-padded with the pad id 0
(reference data/dataloader.py:218-236), or `caption_encodings` f32 [B][768] in the frozen-sentence-embedding mode
(reference model.py:48-50). The LMDB / albumentations stack (reference data/readers.py, data/transforms.py) is CPU-side I/O outside the accelerated path and
its dependencies are absent from the image.

`RandomDataset` is the counterpart of the reference's synthetic dataset (data/dataloader.py:36-114: random 3x224x224 images and
four fixed captions, len 118000). `JsonCaptionDataset` reads the reference's json record format ({"image": path, "caption": str},
data/mock_data.json; reference data/dataloader.py:115-236, JsonDataset).

Text source: `WordPieceTokenizer(vocab.txt)` is BERT's uncased WordPiece (the reference calls
`BertTokenizer.from_pretrained('bert-base-uncased')(caption, padding=False, truncation=True, max_length=L)`, data/dataloader.py:139-141,
196-202) built on the installed `tokenizers` package from a local vocabulary file (`DATA.TOKENIZER_VOCAB`; the HF hub is not reachable
here, so the file must be supplied). Without a vocabulary the offline default is `hash_tokenize`, a deterministic word -> id hash with the
same [CLS] ... [SEP] framing.

Image source: when a record's image file exists it is decoded with PIL and put through the reference's transforms by name
(`DATA.IMAGE_TRANSFORM_{TRAIN,VAL}`; factories.py:112-160): smallest_resize (shorter side -> 256 for DEFAULT_IMAGE_TRANSFORM, else the
crop size), center_crop, random_resized_crop (scale 0.2-1), horizontal_flip (which also swaps "left" / "right" in the caption), color_jitter,
normalize (ImageNet mean / std on [0, 1] pixels), HWC -> CHW float32
(data/dataloader.py:186-192). A record whose file is missing gets a seeded synthetic image (the mock json of the reference points at
files that exist on no machine we have).
"""
import hashlib
import json
import math
import os
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


class WordPieceTokenizer:
    """BERT uncased WordPiece over a local vocab.txt: BertNormalizer (clean text, lower-case, strip accents, CJK spacing) ->
    BertPreTokenizer (whitespace + punctuation) -> greedy longest-match WordPiece ("##" continuation, [UNK] for unmatched words, words
    over 100 characters -> [UNK]) -> "[CLS] ... [SEP]" -> truncation to max_length INCLUDING the two specials — i.e. what the reference's
    `BertTokenizer(caption, padding=False, truncation=True, max_length=L)` returns in `input_ids` (data/dataloader.py:196-202)."""

    def __init__(self, vocab_path: str):
        from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
        if not os.path.isfile(vocab_path):
            raise FileNotFoundError(f"WordPiece vocabulary {vocab_path!r} not found (DATA.TOKENIZER_VOCAB)")
        self.vocab_path = vocab_path
        vocab = {}
        with open(vocab_path, encoding="utf-8") as fh:
            for i, line in enumerate(fh):
                tok = line.rstrip("\n")
                if tok != "" and tok not in vocab:
                    vocab[tok] = i
        for special in ("[PAD]", "[UNK]", "[CLS]", "[SEP]"):
            if special not in vocab:
                raise ValueError(f"{vocab_path}: vocabulary lacks {special}")
        self.pad_token_id, self.cls_token_id, self.sep_token_id = vocab["[PAD]"], vocab["[CLS]"], vocab["[SEP]"]
        self.vocab_size = max(vocab.values()) + 1
        tk = Tokenizer(models.WordPiece(vocab, unk_token="[UNK]", max_input_chars_per_word=100))
        tk.normalizer = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None, lowercase=True)
        tk.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
        tk.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", special_tokens=[("[CLS]", self.cls_token_id), ("[SEP]", self.sep_token_id)])
        self._tk = tk

    def __getstate__(self):          # picklable for DataLoader workers (the reference's tokenizers do the same, data/tokenizers.py:80-92)
        return {"vocab_path": self.vocab_path}

    def __setstate__(self, st):
        self.__init__(st["vocab_path"])

    def __call__(self, caption: str, max_length: int):
        self._tk.enable_truncation(max_length=max_length)
        return self._tk.encode(caption).ids


IMAGENET_COLOR_MEAN = (0.485, 0.456, 0.406)      # reference data/transforms.py:232-235
IMAGENET_COLOR_STD = (0.229, 0.224, 0.225)
DEFAULT_IMAGE_TRANSFORM = ("smallest_resize::256", "center_crop", "normalize")     # reference data/transforms.py:238-244


def swap_left_right(caption: str) -> str:
    """What the reference's HorizontalFlip does to the caption of a flipped image (data/transforms.py:175-181): "left" <-> "right"."""
    return caption.replace("left", "[TMP]").replace("right", "left").replace("[TMP]", "right")


def _transform_args(arg: str, crop_size: int):
    """The part after "::" of a transform name. The reference's syntax is a kwargs dict, `random_resized_crop::{'scale': (0.08, 1.0)}`
    (factories.py:112-115,163-173: eval of the text; here ast.literal_eval — literals only); a bare integer is this package's short form for
    the size. Returns (size, kwargs)."""
    if not arg:
        return crop_size, {}
    import ast
    val = ast.literal_eval(arg)
    if isinstance(val, dict):
        kw = dict(val)
        size = int(kw.pop("size", kw.pop("max_size", crop_size)))
        return size, kw
    return int(val), {}


def _color_jitter(img, rnd, brightness=0.4, contrast=0.4, saturation=0.4, hue=0.1, p=0.8):
    """The reference's `color_jitter` (factories.py:132-134: albumentations ColorJitter(0.4, 0.4, 0.4, 0.1, p = 0.8)) on a PIL image:
    with probability p, brightness / contrast / saturation factors uniform in [1 - x, 1 + x] and a hue shift uniform in [-hue, hue] (fraction of
    the colour circle), applied in a random order. Same distribution family as albumentations / torchvision; the draws come from this
    package's generator, so individual images differ from an albumentations run with the same seed."""
    from PIL import Image, ImageEnhance
    import numpy as np
    if rnd() >= p:
        return img
    fb, fc, fs = (1.0 + (2.0 * rnd() - 1.0) * x for x in (brightness, contrast, saturation))
    fh = (2.0 * rnd() - 1.0) * hue
    order = sorted(range(4), key=lambda _: rnd())
    for op in order:
        if op == 0:
            img = ImageEnhance.Brightness(img).enhance(fb)
        elif op == 1:
            img = ImageEnhance.Contrast(img).enhance(fc)
        elif op == 2:
            img = ImageEnhance.Color(img).enhance(fs)
        else:
            hsv = np.asarray(img.convert("HSV")).copy()
            hsv[..., 0] = (hsv[..., 0].astype(np.int16) + int(round(fh * 255))) % 256
            img = Image.fromarray(hsv, "HSV").convert("RGB")
    return img


def load_image(path: str, transforms, crop_size: int, generator=None, return_flipped=False):
    """PIL decode -> RGB -> the named transforms -> f32 CHW (reference data/dataloader.py:186-192 with the transform table of
    factories.py:112-160). Names may carry arguments as "name::{kwargs dict}" (the reference's syntax, factories.py:112-115) or "name::size";
    factories.py:213-221 passes the crop size to the resize / crop transforms. Random transforms draw from `generator` (a torch.Generator) so
    that a dataset index is reproducible. return_flipped: also return whether `horizontal_flip` fired — the reference's flip swaps "left" and
    "right" in the caption of a flipped image (data/transforms.py:156-181), which the dataset applies before tokenising."""
    from PIL import Image
    import numpy as np
    img = Image.open(path).convert("RGB")

    def rnd():
        return float(torch.rand((), generator=generator))

    normalized, flipped = False, False
    for spec in transforms:
        name, _, arg = spec.partition("::")
        size, kw = _transform_args(arg, crop_size)
        if name == "smallest_resize":            # albumentations SmallestMaxSize: shorter side -> size, aspect kept, bilinear
            w, h = img.size
            sc = size / min(w, h)
            img = img.resize((max(1, round(w * sc)), max(1, round(h * sc))), Image.BILINEAR)
        elif name == "global_resize":
            img = img.resize((size, size), Image.BILINEAR)
        elif name == "center_crop":              # reference data/transforms.py CenterSquareCrop
            w, h = img.size
            l, t = (w - size) // 2, (h - size) // 2
            img = img.crop((l, t, l + size, t + size))
        elif name == "random_resized_crop":      # reference factories.py:124-126: RandomResizedSquareCrop(scale=(0.2, 1.0), ratio=(0.75, 1.333)); 10 tries then centre
            s_lo, s_hi = kw.get("scale", (0.2, 1.0))
            r_lo, r_hi = kw.get("ratio", (0.75, 1.333))
            w, h = img.size
            box = None
            for _ in range(10):
                area = w * h * (s_lo + (s_hi - s_lo) * rnd())
                logr = math.log(r_lo) + (math.log(r_hi) - math.log(r_lo)) * rnd()
                ar = math.exp(logr)
                cw, ch = int(round(math.sqrt(area * ar))), int(round(math.sqrt(area / ar)))
                if 0 < cw <= w and 0 < ch <= h:
                    l, t = int(rnd() * (w - cw + 1)), int(rnd() * (h - ch + 1))
                    box = (l, t, l + cw, t + ch)
                    break
            if box is None:
                s_ = min(w, h)
                box = ((w - s_) // 2, (h - s_) // 2, (w - s_) // 2 + s_, (h - s_) // 2 + s_)
            img = img.crop(box).resize((size, size), Image.BILINEAR)
        elif name == "horizontal_flip":          # reference factories.py:141: p = 0.5; the caption side is the caller's (return_flipped)
            if rnd() < kw.get("p", 0.5):
                img = img.transpose(Image.FLIP_LEFT_RIGHT)
                flipped = not flipped
        elif name in ("color_jitter", "color_jitter8"):
            x = 0.8 if name == "color_jitter8" else 0.4
            img = _color_jitter(img, rnd, kw.get("brightness", x), kw.get("contrast", x), kw.get("saturation", x), kw.get("hue", 0.1), kw.get("p", 0.8))
        elif name == "normalize":
            normalized = True
        else:
            raise KeyError(f"unknown image transform {spec!r}")
    x = torch.from_numpy(np.asarray(img, dtype=np.float32).copy()) / 255.0          # HWC in [0, 1]
    if normalized:                               # albumentations Normalize(mean, std, max_pixel_value=255)
        x = (x - torch.tensor(IMAGENET_COLOR_MEAN)) / torch.tensor(IMAGENET_COLOR_STD)
    x = x.permute(2, 0, 1).contiguous()
    return (x, flipped) if return_flipped else x


class _CaptionDataset(Dataset):
    def __init__(self, mode: str, image_size: int, max_caption_length: int, length: int, seed: int = 0, tokenizer_vocab: str = "",
                 image_transform=DEFAULT_IMAGE_TRANSFORM):
        self.mode, self.image_size, self.max_len, self.length, self.seed = mode, image_size, max_caption_length, length, seed
        self.tokenizer = WordPieceTokenizer(tokenizer_vocab) if tokenizer_vocab else None
        self.image_transform = tuple(image_transform)

    def tokenize(self, caption: str):
        if self.tokenizer is not None:            # the reference's order: NormalizeCaption, then the BERT tokenizer (data/dataloader.py:194-202)
            return self.tokenizer(normalize_caption(caption, self.max_len), self.max_len)
        return hash_tokenize(caption, self.max_len)

    def image_path(self, idx):
        return None

    def __len__(self):
        return self.length

    def caption(self, idx):
        raise NotImplementedError

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        path = self.image_path(idx)
        flipped = False
        if path is not None and os.path.isfile(path):
            image, flipped = load_image(path, self.image_transform, self.image_size, g, return_flipped=True)
        else:
            image = torch.randn(3, self.image_size, self.image_size, generator=g)
        item = {"image_id": torch.tensor(idx, dtype=torch.long), "image": image}
        if self.mode == "sbert":
            item["caption_encodings"] = torch.randn(768, generator=g)
        else:
            caption = self.caption(idx)
            if flipped:          # the image transforms run on (image, RAW caption) pairs in the reference: a flipped image swaps left / right
                caption = swap_left_right(caption)
            item["caption_tokens"] = torch.tensor(self.tokenize(caption), dtype=torch.long)
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
    def __init__(self, mode="train_sbert", image_size=224, max_caption_length=30, length=118000, seed=0, tokenizer_vocab="",
                 image_transform=DEFAULT_IMAGE_TRANSFORM):
        super().__init__(mode, image_size, max_caption_length, length, seed, tokenizer_vocab, image_transform)

    def caption(self, idx):
        return CAPTIONS[idx % len(CAPTIONS)]


class JsonCaptionDataset(_CaptionDataset):
    def __init__(self, json_files, mode="train_sbert", image_size=224, max_caption_length=30, seed=0, tokenizer_vocab="",
                 image_transform=DEFAULT_IMAGE_TRANSFORM, data_root=""):
        self.records = []
        for f in json_files:
            with open(f) as fh:
                self.records += json.load(fh)
        self.data_root = data_root
        super().__init__(mode, image_size, max_caption_length, len(self.records), seed, tokenizer_vocab, image_transform)

    def image_path(self, idx):
        path = self.records[idx].get("image")
        if not path:
            return None
        return path if os.path.isabs(path) or not self.data_root else os.path.join(self.data_root, path)

    def caption(self, idx):
        c = self.records[idx]["caption"]
        return c[0] if isinstance(c, list) else c

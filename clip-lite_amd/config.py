"""Configuration tree with the reference's keys, defaults and semantics (reference config.py:40-259): defaults, then a YAML
file, then a flat ``KEY VALUE ...`` override list, then the derived ``RUN_ID``; the result is frozen. The reference builds
this on fvcore's CfgNode (absent here); this is a small PyYAML-based node with the same attribute access, merge order,
type-checked overrides and ``dump``. The YAMLs under the reference's configs/done/ load unchanged."""
import ast
import copy
from typing import Any, List, Optional

import yaml


class Node(dict):
    """dict with attribute access that can be frozen; nested dicts become Nodes."""

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = Node(v) if isinstance(v, dict) and not isinstance(v, Node) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"Attempted to set {k} to {v}, but the config is immutable")
        self[k] = v

    def freeze(self, flag=True):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, Node):
                v.freeze(flag)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Node) else copy.deepcopy(v)) for k, v in self.items()}

    # -- merging ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _coerce(new, old, key):
        """Accept the replacement only if its type is compatible with the default's (fvcore/yacs behaviour)."""
        if old is None or isinstance(new, type(old)):
            return new
        if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            return float(new)
        if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
            return type(old)(new)
        if isinstance(old, str) and not isinstance(new, (dict, list)):
            return str(new)
        raise ValueError(f"Type mismatch for config key {key}: default {type(old).__name__}, got {type(new).__name__} ({new!r})")

    def merge_from_dict(self, other, prefix=""):
        for k, v in other.items():
            full = prefix + k
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            if isinstance(self[k], Node):
                if not isinstance(v, dict):
                    raise ValueError(f"Config key {full} is a section")
                self[k].merge_from_dict(v, full + ".")
            else:
                self[k] = self._coerce(v, self[k], full)

    def merge_from_file(self, path):
        with open(path, "r") as f:
            self.merge_from_dict(yaml.safe_load(f) or {})

    def merge_from_list(self, lst):
        if len(lst) % 2 != 0:
            raise ValueError(f"Override list has odd length: {lst}; it must be a list of pairs")
        for full, raw in zip(lst[0::2], lst[1::2]):
            node = self
            parts = full.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f"Non-existent config key: {full}")
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent config key: {full}")
            val = raw
            if isinstance(raw, str):
                try:
                    val = ast.literal_eval(raw)
                except (ValueError, SyntaxError):
                    val = raw
            node[parts[-1]] = self._coerce(val, node[parts[-1]], full)


class Config(object):
    def __init__(self, config_file: Optional[str] = None, override_list: List[Any] = []):
        _C = Node()
        _C.RANDOM_SEED = 0
        _C.AMP = True                      # reference: fp16 autocast; here: bf16 storage + bf16 MFMA kernels (False: exact f32 kernels)
        _C.CUDNN_DETERMINISTIC = False     # kept for YAML compatibility; no effect (no cuDNN/MIOpen on this path)
        _C.CUDNN_BENCHMARK = True

        _C.DATA = Node()
        _C.DATA.NAME = "train_sbert"
        _C.DATA.ROOT = "/bigtemp/as3ek/p/vlinfo/datasets/serialized2/"
        _C.DATA.IMAGE_CROP_SIZE = 224
        _C.DATA.MAX_CAPTION_LENGTH = 30
        # (not a reference key) path of a BERT WordPiece vocab.txt; "" = the offline hash tokenizer. The reference downloads
        # 'bert-base-uncased' from the HF hub (data/dataloader.py:139-141), which an air-gapped machine cannot.
        _C.DATA.TOKENIZER_VOCAB = ""
        _C.DATA.USE_SINGLE_CAPTION = False
        _C.DATA.USE_PERCENTAGE = 100.0
        _C.DATA.IMAGE_TRANSFORM_TRAIN = ["random_resized_crop", "horizontal_flip", "color_jitter", "normalize"]
        _C.DATA.IMAGE_TRANSFORM_VAL = ["smallest_resize", "center_crop", "normalize"]
        _C.DATA.JSON_FILES_TRAIN = [
            "/export/share/junnan-li/ALBEF/data/coco_karpathy_train.json",
            "/export/share/junnan-li/ALBEF/data/vg_caption.json",
            "/export/share/junnan-li/ALBEF/data/conceptual_caption_train.json",
            "/export/share/junnan-li/ALBEF/data/conceptual_caption_val.json",
            "/export/share/junnan-li/ALBEF/data/sbu_caption.json",
        ]
        _C.DATA.JSON_FILES_VAL = ["/export/share/junnan-li/ALBEF/data/coco_karpathy_val.json"]
        _C.DATA.NEGATIVE_SAMPLING = "normal"
        _C.DATA.NEGATIVE_SAMPLING_START_ITERATION = 250000
        _C.DATA.CLUSTER_PATH = ""
        _C.DATA.COCO_ROOT = "/bigtemp/as3ek/p/vlinfo/datasets/coco/"

        _C.MODEL = Node()
        _C.MODEL.NAME = "captions"
        _C.MODEL.VISUAL = Node()
        _C.MODEL.VISUAL.NETWORK_NAME = "resnet50"
        _C.MODEL.VISUAL.FEATURE_SIZE = 2048
        _C.MODEL.VISUAL.FROZEN = False
        _C.MODEL.VISUAL.SELF_SUPERVISED = False
        _C.MODEL.TEXTUAL = Node()
        _C.MODEL.TEXTUAL.NAME = "train_sbert"
        _C.MODEL.TEXTUAL.PRETRAINED = False
        _C.MODEL.TEXTUAL.NETWORK_NAME = "bert-base-uncased"
        _C.MODEL.TEXTUAL.WORD_DICT_PATH = "/u/as3ek/github/vlinfo/data/datasets/vocab/word_dict.json"
        _C.MODEL.TEXTUAL.LOAD_GLOVE = False
        _C.MODEL.TEXTUAL.GLOVE_PATH = "/u/as3ek/github/vlinfo/data/datasets/glove/glove.42B.300d.txt"
        _C.MODEL.TEXTUAL.TRAIN_EMBEDDINGS = False
        _C.MODEL.TEXTUAL.TRANSFORM = False
        _C.MODEL.TEXTUAL.FEATURE_SIZE = 768
        _C.MODEL.TEXTUAL.SELF_SUPERVISED = False
        _C.MODEL.TEXTUAL.NUM_HIDDEN_LAYERS = 12
        _C.MODEL.LOSS = Node()
        _C.MODEL.LOSS.NAME = "jsd"
        _C.MODEL.LOSS.TYPE = "dot"
        _C.MODEL.LOSS.IMAGE_PRIOR = True
        _C.MODEL.LOSS.TEXT_PRIOR = True
        _C.MODEL.LOSS.PRIOR_WEIGHT = 0.1

        _C.OPTIM = Node()
        _C.OPTIM.OPTIMIZER_NAME = "sgd"
        _C.OPTIM.SGD_MOMENTUM = 0.9
        _C.OPTIM.WEIGHT_DECAY = 0.0001
        _C.OPTIM.NO_DECAY = ".*textual.(embedding|transformer).*(norm.*|bias)"
        _C.OPTIM.CLIP_GRAD_NORM = 10.0
        _C.OPTIM.LOOKAHEAD = Node()
        _C.OPTIM.LOOKAHEAD.USE = True
        _C.OPTIM.LOOKAHEAD.ALPHA = 0.5
        _C.OPTIM.LOOKAHEAD.STEPS = 5
        _C.OPTIM.BATCH_SIZE = 256
        _C.OPTIM.CNN_LR = 0.2
        _C.OPTIM.LR = 0.001
        _C.OPTIM.TRANS_LR = 0.001
        _C.OPTIM.MIN_LR_MULT = 0.0
        _C.OPTIM.NUM_ITERATIONS = 500000
        _C.OPTIM.WARMUP_STEPS = 10000
        _C.OPTIM.LR_DECAY_NAME = "cosine"
        _C.OPTIM.LR_STEPS = []
        _C.OPTIM.LR_GAMMA = 0.1

        _C.RUN_ID = ""

        self._C = _C
        if config_file is not None:
            self._C.merge_from_file(config_file)
        self._C.merge_from_list(list(override_list))
        self.add_derived_params()
        self._C.freeze()

    def add_derived_params(self):
        """RUN_ID doubles as the checkpoint sub-directory name (reference config.py:223-250); field order kept."""
        c = self._C
        fields = [("V", c.MODEL.VISUAL.NETWORK_NAME), ("T", c.MODEL.TEXTUAL.NAME), ("Ty", c.MODEL.LOSS.TYPE),
                  ("Vs", c.MODEL.VISUAL.SELF_SUPERVISED), ("Ts", c.MODEL.TEXTUAL.SELF_SUPERVISED), ("N", c.DATA.NEGATIVE_SAMPLING),
                  ("B", c.OPTIM.BATCH_SIZE), ("O", c.OPTIM.OPTIMIZER_NAME), ("B", c.OPTIM.BATCH_SIZE), ("D", c.OPTIM.LR_DECAY_NAME),
                  ("Ni", c.OPTIM.NUM_ITERATIONS), ("ID", c.RUN_ID)]
        self._C.RUN_ID = "/" + "_".join(f"{k}?{v}" for k, v in fields)

    def dump(self, file_path: str):
        with open(file_path, "w") as f:
            yaml.safe_dump(self._C.to_dict(), f, default_flow_style=False)

    def __getattr__(self, attr: str):
        return self._C.__getattr__(attr)

    def __str__(self):
        return yaml.safe_dump(self._C.to_dict(), default_flow_style=False)

    def __repr__(self):
        return repr(self._C.to_dict())

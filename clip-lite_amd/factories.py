"""Name -> class registries with the reference's factory interface (reference factories.py:36-73 `Factory.create/from_config`,
:303-434 model factories, :437-487 OptimizerFactory, :490-531 LRSchedulerFactory, :169-238 PretrainingDatasetFactory).
Model/optimizer/scheduler products are the HIP-backed classes of this package; dataset products are the batch sources of
data.py (the albumentations / LMDB-backed datasets of the reference need dependencies and files that are not available)."""
import re
from typing import Any, Callable, Dict, Iterable, List

from torch import nn

from . import data as vdata
from .config import Config
from .encoder import ImageEncoder, TextEncoder
from .loss import InfoNCELoss, JSDInfoMaxLoss
from .model import VLInfoModel
from .optim import FusedAdamW, FusedSGD, Lookahead, lr_scheduler


class Factory(object):
    PRODUCTS: Dict[str, Callable] = {}

    def __init__(self):
        raise ValueError(f"Cannot instantiate {self.__class__.__name__} object, use `create` classmethod.")

    @classmethod
    def create(cls, name: str, *args, **kwargs) -> Any:
        if name not in cls.PRODUCTS:
            raise KeyError(f"{cls.__class__.__name__} cannot create {name}.")
        return cls.PRODUCTS[name](*args, **kwargs)

    @classmethod
    def from_config(cls, config: Config) -> Any:
        raise NotImplementedError


class PretrainingDatasetFactory(Factory):
    PRODUCTS: Dict[str, Callable] = {"random": vdata.RandomDataset, "json": vdata.JsonCaptionDataset}

    @classmethod
    def from_config(cls, config: Config, split: str = "train"):
        _C = config
        kwargs = {"mode": _C.MODEL.TEXTUAL.NAME, "image_size": _C.DATA.IMAGE_CROP_SIZE, "max_caption_length": _C.DATA.MAX_CAPTION_LENGTH,
                  "tokenizer_vocab": _C.DATA.TOKENIZER_VOCAB,
                  # transform names as configured; resize / crop transforms take IMAGE_CROP_SIZE (reference factories.py:213-224)
                  "image_transform": tuple(getattr(_C.DATA, f"IMAGE_TRANSFORM_{split.upper()}"))}
        if _C.MODEL.NAME == "json":
            kwargs["json_files"] = list(_C.DATA.JSON_FILES_TRAIN if split == "train" else _C.DATA.JSON_FILES_VAL)
            kwargs["data_root"] = _C.DATA.ROOT
        elif _C.MODEL.NAME == "random":
            kwargs["length"] = 118000 if split == "train" else 5000
        else:
            raise KeyError(f"MODEL.NAME={_C.MODEL.NAME!r}: the LMDB-backed COCO dataset needs lmdb/albumentations and the serialized "
                           "dataset, which are not available; use MODEL.NAME random or json")
        return cls.create(_C.MODEL.NAME, **kwargs)


class VisualBackboneFactory(Factory):
    PRODUCTS: Dict[str, Callable] = {"captions": ImageEncoder, "random": ImageEncoder, "json": ImageEncoder}

    @classmethod
    def from_config(cls, config: Config) -> ImageEncoder:
        # like the reference (factories.py:324-327) only the network name is forwarded
        return cls.create(config.MODEL.NAME, img_enc_net=config.MODEL.VISUAL.NETWORK_NAME)


class TextualHeadFactory(Factory):
    PRODUCTS: Dict[str, Callable] = {"glove": TextEncoder, "sbert": TextEncoder, "train_sbert": TextEncoder}

    @classmethod
    def from_config(cls, config: Config) -> nn.Module:
        _C = config
        name = _C.MODEL.TEXTUAL.NAME
        kwargs = {
            "word_dict": {}, "mode": name, "transform_embedding": _C.MODEL.TEXTUAL.TRANSFORM, "txt_enc_dim": _C.MODEL.TEXTUAL.FEATURE_SIZE,
            "load_glove": _C.MODEL.TEXTUAL.LOAD_GLOVE, "glove_path": _C.MODEL.TEXTUAL.GLOVE_PATH, "train_enc": _C.MODEL.TEXTUAL.TRAIN_EMBEDDINGS,
            "pretrained": _C.MODEL.TEXTUAL.PRETRAINED, "model_name": _C.MODEL.TEXTUAL.NETWORK_NAME,
            "num_hidden_layers": _C.MODEL.TEXTUAL.NUM_HIDDEN_LAYERS,
        }
        return cls.create(name, **kwargs)


class LossFactory(Factory):
    PRODUCTS: Dict[str, Callable] = {"jsd": JSDInfoMaxLoss, "infonce": InfoNCELoss}     # "infonce": BASELINE config 4 (not in the reference)

    @classmethod
    def from_config(cls, config: Config) -> JSDInfoMaxLoss:
        _C = config
        kwargs = {
            "image_dim": _C.MODEL.VISUAL.FEATURE_SIZE, "text_dim": _C.MODEL.TEXTUAL.FEATURE_SIZE, "type": _C.MODEL.LOSS.TYPE,
            "image_prior": _C.MODEL.LOSS.IMAGE_PRIOR, "text_prior": _C.MODEL.LOSS.TEXT_PRIOR, "prior_weight": _C.MODEL.LOSS.PRIOR_WEIGHT,
            "visual_self_supervised": _C.MODEL.VISUAL.SELF_SUPERVISED, "textual_self_supervised": _C.MODEL.TEXTUAL.SELF_SUPERVISED,
        }
        return cls.create(_C.MODEL.LOSS.NAME, **kwargs)


class PretrainingModelFactory(Factory):
    PRODUCTS: Dict[str, Callable] = {"captions": VLInfoModel, "random": VLInfoModel, "json": VLInfoModel}

    @classmethod
    def from_config(cls, config: Config) -> nn.Module:
        _C = config
        visual = VisualBackboneFactory.from_config(_C)      # construction order as in the reference (factories.py:429-431)
        textual = TextualHeadFactory.from_config(_C)
        loss = LossFactory.from_config(_C)
        return cls.create(_C.MODEL.NAME, textual, visual, loss, _C.MODEL.TEXTUAL.NAME, _C.AMP)


class OptimizerFactory(Factory):
    PRODUCTS: Dict[str, Callable] = {"sgd": FusedSGD, "adamw": FusedAdamW}

    @classmethod
    def from_config(cls, config: Config, named_parameters: Iterable[Any]):
        _C = config
        param_groups: List[Dict[str, Any]] = []
        for name, param in named_parameters:
            wd = 0.0 if re.match(_C.OPTIM.NO_DECAY, name) else _C.OPTIM.WEIGHT_DECAY
            if "image_encoder" in name:
                lr = _C.OPTIM.CNN_LR
            elif "text_encoder" in name:
                lr = _C.OPTIM.TRANS_LR
            else:
                lr = _C.OPTIM.LR
            param_groups.append({"params": [param], "lr": lr, "weight_decay": wd})
        kwargs = {"momentum": _C.OPTIM.SGD_MOMENTUM} if _C.OPTIM.OPTIMIZER_NAME == "sgd" else {}          # (reference factories.py:478-483)
        optimizer = cls.create(_C.OPTIM.OPTIMIZER_NAME, param_groups, **kwargs)
        if _C.OPTIM.LOOKAHEAD.USE:
            optimizer = Lookahead(optimizer, k=_C.OPTIM.LOOKAHEAD.STEPS, alpha=_C.OPTIM.LOOKAHEAD.ALPHA)
        return optimizer


class LRSchedulerFactory(Factory):
    PRODUCTS: Dict[str, Callable] = {
        "none": lr_scheduler.LinearWarmupNoDecayLR, "multistep": lr_scheduler.LinearWarmupMultiStepLR,
        "linear": lr_scheduler.LinearWarmupLinearDecayLR, "cosine": lr_scheduler.LinearWarmupCosineAnnealingLR,
    }

    @classmethod
    def from_config(cls, config: Config, optimizer):
        _C = config
        kwargs = {"total_steps": _C.OPTIM.NUM_ITERATIONS, "warmup_steps": _C.OPTIM.WARMUP_STEPS}
        if _C.OPTIM.LR_DECAY_NAME == "multistep":
            kwargs.update(gamma=_C.OPTIM.LR_GAMMA, milestones=_C.OPTIM.LR_STEPS)
        if _C.OPTIM.LR_DECAY_NAME == "cosine":
            kwargs.update(min_mult=_C.OPTIM.MIN_LR_MULT)
        return cls.create(_C.OPTIM.LR_DECAY_NAME, optimizer, **kwargs)

"""VLInfoModel — the reference's model wrapper (reference model.py:15-113) over the HIP encoders and loss.

`is_amp=True` (the reference's autocast/fp16 mode) selects bf16 storage + bf16 MFMA kernels; `is_amp=False` (the reference's
fp32 mode, also what its CPU path computes) selects the exact-f32 kernels (v_mfma_f32_32x32x2_f32), which is the parity mode.
Moving the model to a GPU (`.to(device)`, as reference train.py:136 does) builds the flat parameter arena on that device.
"""
import torch
from torch import nn

from . import hip
from .encoder import ImageEncoder, TextEncoder
from .loss import JSDInfoMaxLoss
from .runtime import DeviceRuntime


def attach_runtime(model, device, lowp, seed=0):
    groups = []
    te = getattr(model, "text_encoder", None)
    if te is not None and hasattr(te, "strans"):
        groups = te.strans.contiguous_groups("text_encoder.strans.")
    rt = DeviceRuntime(model, device, lowp, groups, seed)
    # a leaf that requires grad: threads the custom autograd Functions into the graph even when their tensor inputs
    # (images, token ids) do not require grad; parameter gradients are written straight into the arena by the kernels
    rt.anchor = torch.zeros(1, device=rt.device, requires_grad=True)
    return rt


class VLInfoModel(nn.Module):
    def __init__(self, text_encoder: TextEncoder, image_encoder: ImageEncoder, loss: JSDInfoMaxLoss, mode: str = "sbert", is_amp: bool = True):
        super().__init__()
        self.text_encoder = text_encoder
        self.image_encoder = image_encoder
        self.loss = loss
        self.mode = mode
        self.is_amp = is_amp
        self.overlap_encoders = True     # text encoder on a second stream, concurrent with the image encoder
        self._rt = None

    # -- device placement builds the arena ---------------------------------------------------------------------------
    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        p = next(self.parameters(), None)
        dev = p.device if p is not None else None
        if dev is not None and (dev.type == "cuda" or hip._allow_host_tensors):
            if self._rt is None or self._rt.device != dev:
                self._rt = attach_runtime(self, dev, self.is_amp, seed=torch.initial_seed() % (2 ** 31))
        return out

    @property
    def runtime(self):
        if self._rt is None:
            raise RuntimeError("clip_lite_amd: move the model to a GPU first (model.to(device)); there is no CPU path")
        return self._rt

    def state_dict(self, *args, **kw):
        if self._rt is not None:
            self._rt.arena.flush_pending()          # a deferred share of the last update (TrainStep defer_update) lands before parameters are read
        return super().state_dict(*args, **kw)

    def load_state_dict(self, state_dict, strict=True, **kw):
        if self._rt is not None:
            self._rt.arena.flush_pending()
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        if self._rt is not None:
            self._rt.arena.refresh_lowp()
        return out

    def forward(self, batch):
        rt = self.runtime
        if not self.training:
            rt.arena.flush_pending()          # an eval forward between deferred steps must see the completed update
        if self.mode == "sbert":
            image_features = self.image_encoder(batch["image"])
            text_features = self.text_encoder(batch["caption_encodings"])
        elif self.mode == "train_sbert":
            extra = any(k in batch for k in ("neg_input_ids", "aug_image", "aug_input_ids"))
            ex = getattr(rt, "exchange", None)
            if ex is not None:
                # with hard negatives / augmented views each encoder runs several times per step, so "this module's gradients are final"
                # only holds after the last backward: the exchange then reduces the whole arena at the end of the step instead
                ex.defer = extra
            if extra:
                return self._forward_with_extras(batch)
            # The two encoders are independent until the loss: the text encoder is enqueued on a second HIP stream so its
            # (small-grid, latency-bound) BERT kernels share the chip with the ResNet's. Autograd replays each backward on the
            # stream its forward ran on, so the two backward passes overlap the same way; under graph capture the fork/join
            # becomes two parallel branches of the hipGraph.
            text_in = {"input_ids": batch["input_ids"], "attention_mask": batch["attention_mask"]}
            if self.overlap_encoders and rt.device.type == "cuda":
                if rt.side_stream is None:
                    rt.side_stream = rt.new_side_stream()
                main = rt.main_stream = torch.cuda.current_stream(rt.device)
                rt.side_stream.wait_stream(main)
                with torch.cuda.stream(rt.side_stream):
                    text_features = self.text_encoder(text_in)
                image_features = self.image_encoder(batch["image"])
                main.wait_stream(rt.side_stream)
            else:
                image_features = self.image_encoder(batch["image"])
                text_features = self.text_encoder(text_in)
        else:
            raise NotImplementedError(f"mode {self.mode!r}")
        loss_dict = self.loss(image_features=image_features, text_features=text_features)
        return self._output(loss_dict)

    def _forward_with_extras(self, batch):
        """reference model.py:61-92: hard negatives from the clustered dataset (`neg_image`, `neg_input_ids`, `neg_attention_mask`) and
        augmented views (`aug_image`; `aug_input_ids`, `aug_attention_mask`) go through the same encoders — separate calls, so separate
        BatchNorm batch statistics and running-stat updates per call, as in the reference — and on into the loss."""
        enc_t = lambda ids, mask: self.text_encoder({"input_ids": batch[ids], "attention_mask": batch[mask]})
        image_features = self.image_encoder(batch["image"])
        text_features = enc_t("input_ids", "attention_mask")
        neg_image = neg_text = aug_image = aug_text = None
        if "neg_input_ids" in batch:
            neg_image = self.image_encoder(batch["neg_image"])
            neg_text = enc_t("neg_input_ids", "neg_attention_mask")
        if "aug_image" in batch:
            aug_image = self.image_encoder(batch["aug_image"])
        if "aug_input_ids" in batch:
            aug_text = enc_t("aug_input_ids", "aug_attention_mask")
        loss_dict = self.loss(image_features=image_features, text_features=text_features, neg_image_features=neg_image, neg_text_features=neg_text,
                              aug_image_features=aug_image, aug_text_features=aug_text)
        return self._output(loss_dict)

    @staticmethod
    def _output(loss_dict):
        return {
            "loss": loss_dict["total_loss"],
            "loss_components": {
                "total_loss": loss_dict["total_loss"].clone().detach(),
                "cross_modal_loss": loss_dict["cross_modal_loss"].clone().detach(),
                "visual_loss": loss_dict["visual_loss"].clone().detach(),
                "textual_loss": loss_dict["textual_loss"].clone().detach(),
            },
        }

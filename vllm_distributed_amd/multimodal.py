"""Image inputs for vision-language models (LLaVA family).

A request may carry images (``{"prompt": ..., "multi_modal_data": {"image": <PIL image | path | data URL | list>}}``
offline; ``image_url`` content parts in ``/v1/chat/completions``).  The front end turns each image into the
pixel tensor the vision tower reads, and every image placeholder token of the prompt into as many copies as the
tower yields embeddings for one image; the positions of those copies travel with the request
(``MultiModalInput.offset`` / ``length``), and the workers overwrite the embeddings at those positions with the
projected image features while the prompt is prefilled - in whatever chunks the scheduler cuts it.

The prefix cache keys a block by its tokens; placeholder tokens look the same for every image, so blocks that
overlap an image also hash the image's content (``mm_hash``).

vLLM: ``vllm/multimodal`` + ``vllm/model_executor/models/llava.py``; the reference framework reaches it through
the app it serves (/root/reference/src/launch.py:429).  No network here: ``http(s)://`` image URLs are refused,
``data:`` URLs and local files are read.
"""
from __future__ import annotations

import base64
import hashlib
import io
import json
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass
class MultiModalInput:
    """One image of a request: where its placeholder tokens sit in the prompt, and its pixels."""
    offset: int
    length: int
    pixel_values: np.ndarray            # [3, H, W] float16, normalised
    mm_hash: int


def load_image(src: Any):
    """PIL image from a PIL image, a ``data:`` URL, a ``file://`` URL or a path."""
    from PIL import Image
    if isinstance(src, Image.Image):
        return src
    if isinstance(src, dict) and "url" in src:
        src = src["url"]
    if isinstance(src, (bytes, bytearray)):
        return Image.open(io.BytesIO(bytes(src)))
    if not isinstance(src, str):
        raise ValueError(f"cannot read an image from {type(src).__name__}")
    if src.startswith("data:"):
        head, _, payload = src.partition(",")
        if ";base64" not in head:
            raise ValueError("data: image URLs must be base64-encoded")
        try:
            return Image.open(io.BytesIO(base64.b64decode(payload, validate=False)))
        except Exception as e:
            raise ValueError(f"undecodable image data ({e})") from None
    if src.startswith(("http://", "https://")):
        raise ValueError("remote image URLs are not fetched by this server: send the image as a data: URL")
    path = src[len("file://"):] if src.startswith("file://") else src
    if not os.path.isfile(path):
        raise ValueError(f"no image at {path!r}")
    return Image.open(path)


class ClipImageProcessor:
    """CLIP preprocessing: shortest edge to ``size`` (bicubic), centre crop, scale to [0, 1], normalise."""

    def __init__(self, size: int = 336, crop: Optional[int] = None, mean: Sequence[float] = CLIP_MEAN,
                 std: Sequence[float] = CLIP_STD, do_center_crop: bool = True):
        self.size, self.crop = int(size), int(crop or size)
        self.mean = np.asarray(mean, dtype=np.float32).reshape(3, 1, 1)
        self.std = np.asarray(std, dtype=np.float32).reshape(3, 1, 1)
        self.do_center_crop = do_center_crop

    @classmethod
    def from_model(cls, model_dir: Optional[str], vision_cfg: Dict[str, Any]) -> "ClipImageProcessor":
        side = int(vision_cfg.get("image_size", 224))
        pc: Dict[str, Any] = {}
        if model_dir and os.path.isfile(os.path.join(model_dir, "preprocessor_config.json")):
            with open(os.path.join(model_dir, "preprocessor_config.json")) as f:
                pc = json.load(f)
        size = pc.get("size", side)
        if isinstance(size, dict):
            size = size.get("shortest_edge") or size.get("height") or side
        crop = pc.get("crop_size", side)
        if isinstance(crop, dict):
            crop = crop.get("height") or side
        if int(crop) != side:
            raise ValueError(f"preprocessor crop size {crop} != the vision tower's image size {side}")
        return cls(int(size), int(crop), pc.get("image_mean", CLIP_MEAN), pc.get("image_std", CLIP_STD),
                   bool(pc.get("do_center_crop", True)))

    def __call__(self, image) -> np.ndarray:
        from PIL import Image
        img = image.convert("RGB")
        w, h = img.size
        if self.do_center_crop:
            # short side to ``size``, the long side scaled alike and truncated (what transformers' resize does)
            nw, nh = (self.size, int(self.size * h / w)) if w <= h else (int(self.size * w / h), self.size)
            img = img.resize((nw, nh), Image.BICUBIC)
            left, top = (nw - self.crop) // 2, (nh - self.crop) // 2
            img = img.crop((left, top, left + self.crop, top + self.crop))
        else:
            img = img.resize((self.crop, self.crop), Image.BICUBIC)
        x = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / 255.0
        return ((x - self.mean) / self.std).astype(np.float16)


def image_hash(pixels: np.ndarray) -> int:
    return int.from_bytes(hashlib.blake2b(np.ascontiguousarray(pixels).tobytes(), digest_size=8).digest(), "little")


def tokens_per_image(cfg: Dict[str, Any]) -> int:
    v = cfg["vision_config"]
    n = (int(v.get("image_size", 224)) // int(v.get("patch_size", 32))) ** 2
    return n + (1 if cfg.get("vision_feature_select_strategy", "default") == "full" else 0)


def expand_image_placeholders(ids: Sequence[int], images: List[np.ndarray], image_token: int,
                              per_image: int) -> Tuple[List[int], List[MultiModalInput]]:
    """Every ``image_token`` in ``ids`` becomes ``per_image`` copies; the k-th occurrence belongs to the k-th
    image.  A prompt whose placeholders are already expanded (runs of exactly ``per_image``) is taken as it is."""
    where = [i for i, t in enumerate(ids) if t == image_token]
    if len(where) == len(images) * per_image and per_image > 1 and images:
        runs = [where[k * per_image:(k + 1) * per_image] for k in range(len(images))]
        if all(r[-1] - r[0] == per_image - 1 for r in runs):
            return list(ids), [MultiModalInput(r[0], per_image, px, image_hash(px)) for r, px in zip(runs, images)]
    if len(where) != len(images):
        raise ValueError(f"the prompt has {len(where)} image placeholder(s) for {len(images)} image(s)")
    out: List[int] = []
    items: List[MultiModalInput] = []
    k = 0
    for t in ids:
        if t == image_token:
            items.append(MultiModalInput(len(out), per_image, images[k], image_hash(images[k])))
            out.extend([image_token] * per_image)
            k += 1
        else:
            out.append(t)
    return out, items


def block_mm_salt(items: Optional[List[MultiModalInput]], start: int, end: int) -> Optional[Tuple[int, ...]]:
    """What a KV block over token positions ``[start, end)`` must hash besides its tokens: the images it overlaps."""
    if not items:
        return None
    hit = tuple(m.mm_hash for m in items if m.offset < end and m.offset + m.length > start)
    return hit or None

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
import math
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
    pixel_values: np.ndarray            # [3, H, W] float16, normalised ([1 + tiles, 3, H, W]: any-resolution models)
    mm_hash: int
    image_size: Optional[Tuple[int, int]] = None      # (height, width) of the original image, where the model unpads


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


# ------------------------------------------------------------------ any-resolution tiling (LLaVA-NeXT)
def select_best_resolution(size: Tuple[int, int], candidates: Sequence[Sequence[int]]) -> Tuple[int, int]:
    """The (height, width) among ``candidates`` that keeps most of the image's pixels after an aspect-preserving
    fit, and wastes the least area among those (LLaVA-NeXT's ``image_grid_pinpoints`` rule)."""
    h, w = size
    best, best_fit, best_waste = None, -1, None
    for ch, cw in candidates:
        scale = min(cw / w, ch / h)
        dw, dh = int(w * scale), int(h * scale)
        fit = min(dw * dh, w * h)
        waste = cw * ch - fit
        if fit > best_fit or (fit == best_fit and waste < best_waste):
            best, best_fit, best_waste = (int(ch), int(cw)), fit, waste
    return best


def anyres_unpadded_grid(size: Tuple[int, int], grid: Tuple[int, int]) -> Tuple[int, int]:
    """Rows and columns of a ``grid`` (rows, columns) of patch features that are left once the padding the image
    got on its way to the tiled resolution is cut off again (the model's ``unpad_image``)."""
    oh, ow = size
    ch, cw = grid
    if ow / oh > cw / ch:
        nh = int(round(oh * (cw / ow), 7))
        pad = (ch - nh) // 2
        return ch - 2 * pad, cw
    nw = int(round(ow * (ch / oh), 7))
    pad = (cw - nw) // 2
    return ch, cw - 2 * pad


def anyres_tokens(cfg: Dict[str, Any], size: Tuple[int, int]) -> int:
    """Embeddings one image of ``size`` (height, width) yields: the base view's patches, then the unpadded tile
    grid with one ``image_newline`` per row."""
    v = cfg["vision_config"]
    side, per = int(v.get("image_size", 336)), int(v.get("image_size", 336)) // int(v.get("patch_size", 14))
    bh, bw = select_best_resolution(size, cfg["image_grid_pinpoints"])
    rows, cols = anyres_unpadded_grid(size, (bh // side * per, bw // side * per))
    return per * per + rows * (cols + 1)


class AnyResImageProcessor:
    """LLaVA-NeXT preprocessing: the image resized into the best of the model's tiled resolutions (aspect kept,
    padded with black to the full grid) and cut into tiles, behind a base view of the whole image squeezed to one
    tile; every view then gets the CLIP preprocessing.  Returns ``[1 + tiles, 3, S, S]`` float16."""

    def __init__(self, clip: ClipImageProcessor, pinpoints: Sequence[Sequence[int]]):
        self.clip, self.pinpoints = clip, [tuple(int(x) for x in p) for p in pinpoints]

    def __call__(self, image) -> np.ndarray:
        from PIL import Image
        img = image.convert("RGB")
        w, h = img.size
        side = self.clip.crop
        bh, bw = select_best_resolution((h, w), self.pinpoints)
        scale_w, scale_h = bw / w, bh / h
        if scale_w < scale_h:
            nw, nh = bw, min(math.ceil(h * scale_w), bh)
        else:
            nh, nw = bh, min(math.ceil(w * scale_h), bw)
        resized = img.resize((nw, nh), Image.BICUBIC)
        canvas = Image.new("RGB", (bw, bh), (0, 0, 0))
        canvas.paste(resized, ((bw - nw) // 2, (bh - nh) // 2))
        views = [img.resize((side, side), Image.BICUBIC)]
        for top in range(0, bh, side):
            for left in range(0, bw, side):
                views.append(canvas.crop((left, top, left + side, top + side)))
        return np.stack([self.clip(v) for v in views])


def image_hash(pixels: np.ndarray) -> int:
    return int.from_bytes(hashlib.blake2b(np.ascontiguousarray(pixels).tobytes(), digest_size=8).digest(), "little")


def tokens_per_image(cfg: Dict[str, Any]) -> int:
    v = cfg["vision_config"]
    n = (int(v.get("image_size", 224)) // int(v.get("patch_size", 32))) ** 2
    return n + (1 if cfg.get("vision_feature_select_strategy", "default") == "full" else 0)


def expand_image_placeholders(ids: Sequence[int], images: List[np.ndarray], image_token: int, per_image,
                              sizes: Optional[List[Tuple[int, int]]] = None) -> Tuple[List[int], List[MultiModalInput]]:
    """Every ``image_token`` in ``ids`` becomes as many copies as its image yields embeddings (``per_image``: one
    number, or one per image); the k-th occurrence belongs to the k-th image.  A prompt whose placeholders are
    already expanded (runs of exactly those lengths) is taken as it is."""
    counts = [int(per_image)] * len(images) if isinstance(per_image, int) else [int(c) for c in per_image]
    sizes = sizes or [None] * len(images)
    where = [i for i, t in enumerate(ids) if t == image_token]
    if images and len(where) == sum(counts) and min(counts) > 1:
        runs, at = [], 0
        for c in counts:
            runs.append(where[at:at + c])
            at += c
        if all(r[-1] - r[0] == c - 1 for r, c in zip(runs, counts)):
            return list(ids), [MultiModalInput(r[0], c, px, image_hash(px), sz)
                               for r, c, px, sz in zip(runs, counts, images, sizes)]
    if len(where) != len(images):
        raise ValueError(f"the prompt has {len(where)} image placeholder(s) for {len(images)} image(s)")
    out: List[int] = []
    items: List[MultiModalInput] = []
    k = 0
    for t in ids:
        if t == image_token:
            items.append(MultiModalInput(len(out), counts[k], images[k], image_hash(images[k]), sizes[k]))
            out.extend([image_token] * counts[k])
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

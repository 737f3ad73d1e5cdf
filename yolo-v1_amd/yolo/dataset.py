"""Data side of the reference's surface (src/yolo/dataset.py), without torchvision.

``VOCDetectionYOLO`` / ``CombinedVOCDataset`` / ``create_voc_datasets`` keep the reference's constructor arguments,
attributes (``S, B, C, class_names, class_to_idx, augment, target_size``) and return types; what the reference delegates
to ``torchvision.datasets.VOCDetection`` and ``torchvision.transforms.v2`` (un-vendored dependencies) is restated here:

  * the VOC directory layout ``<root>/[<Kaggle split dir>/]VOCdevkit/VOC<year>/{JPEGImages, Annotations,
    ImageSets/Main/<image_set>.txt}`` and the XML -> nested-dict conversion of ``VOCDetection.parse_voc_xml`` (the
    annotation dicts passed to ``_extract_bboxes_from_annotation`` have torchvision's shape, dataset.py:411-467);
  * evaluation transform = Resize (PIL bilinear) -> ToTensor -> Normalize, i.e. ``yolo.inference._Preprocess``
    (bit-identical to the reference's v2 pipeline for PIL inputs, tests/test_preprocess_cpu.py);
  * training augmentation = box-aware RandomResizedCrop(scale (0.8, 1.2), ratio (0.8, 1.2)) + ColorJitter(brightness 0.5,
    saturation 0.5, hue 0.1) (dataset.py:288-319), same distributions, drawn from ``torch``'s global RNG -- the random
    STREAM differs from torchvision's, so augmented samples are "parity unpinned" (statistics, not bits);
  * ``download=True`` (kagglehub) is not available offline and raises.

``encode_target`` restates ``_encode_target`` (dataset.py:487-532); ``SyntheticYOLODataset`` feeds benchmarks and tests.
"""

from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from pathlib import Path
from typing import List, Tuple

import numpy as np
import torch
from PIL import Image, ImageEnhance
from torch.utils.data import Dataset

VOC_CLASSES = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
               "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]


def encode_target(bboxes, class_ids, S: int = 7, B: int = 2, C: int = 20) -> torch.Tensor:
    """Boxes (x_center, y_center, w, h in [0,1]) + class ids -> (S, S, 5B+C) target.
    The first object that lands in a cell owns it; only box slot 0 is filled; class is one-hot."""
    t = torch.zeros((S, S, 5 * B + C))
    for (xc, yc, w, h), cid in zip(bboxes, class_ids):
        i = min(int(S * yc), S - 1)
        j = min(int(S * xc), S - 1)
        if t[i, j, 4] == 0:
            t[i, j, 0] = S * xc - j
            t[i, j, 1] = S * yc - i
            t[i, j, 2] = w
            t[i, j, 3] = h
            t[i, j, 4] = 1.0
            t[i, j, 5 * B + cid] = 1.0
    return t


class SyntheticYOLODataset(Dataset):
    """Random 448x448 images ~N(0,1) with 0..max_obj encoded objects (the benchmark input of SURVEY.md 8d)."""

    def __init__(self, length: int = 256, S: int = 7, B: int = 2, C: int = 20, max_obj: int = 3, seed: int = 0, size: int = 448):
        self.length, self.S, self.B, self.C, self.max_obj, self.seed, self.size = length, S, B, C, max_obj, seed, size

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, idx: int):
        rng = np.random.Generator(np.random.PCG64([self.seed, idx]))
        img = torch.from_numpy(rng.standard_normal((3, self.size, self.size), dtype=np.float32))
        k = int(rng.integers(0, self.max_obj + 1))
        boxes = [(*rng.uniform(0, 1, 2), *rng.uniform(0.05, 0.9, 2)) for _ in range(k)]
        cids = [int(rng.integers(0, self.C)) for _ in range(k)]
        return img, encode_target(boxes, cids, self.S, self.B, self.C)


# ------------------------------------------------------------------------------------------------------------------
# PASCAL VOC on disk
# ------------------------------------------------------------------------------------------------------------------
def parse_voc_xml(node: ET.Element) -> dict:
    """XML element -> nested dict with the shape torchvision's ``VOCDetection.parse_voc_xml`` gives: a tag that repeats
    under one parent becomes a list, ``annotation["object"]`` is ALWAYS a list (empty without objects), leaves are the
    stripped text."""
    children = list(node)
    if not children:
        return {node.tag: (node.text or "").strip()}
    grouped: dict = {}
    for child in children:
        for k, v in parse_voc_xml(child).items():
            grouped.setdefault(k, []).append(v)
    inner = {k: (v[0] if len(v) == 1 else v) for k, v in grouped.items()}
    if node.tag == "annotation":
        inner["object"] = grouped.get("object", [])
    return {node.tag: inner}


def _uniform(a: float, b: float) -> float:
    return float(torch.empty(1).uniform_(a, b).item())


class _Augment:
    """RandomResizedCrop(size, scale=(0.8, 1.2), ratio=(0.8, 1.2)) + ColorJitter(brightness 0.5, saturation 0.5, hue 0.1),
    applied to a PIL image and its pixel-space XYXY boxes (torchvision.transforms.v2 semantics: the crop is sampled by
    area fraction x log-uniform aspect ratio, 10 tries, else the central crop at the clamped ratio; boxes are shifted,
    clamped to the crop and scaled with it; the colour operations run in a random order)."""

    def __init__(self, size: Tuple[int, int], scale=(0.8, 1.2), ratio=(0.8, 1.2), brightness=0.5, saturation=0.5, hue=0.1):
        self.size, self.scale, self.ratio = size, scale, ratio
        self.brightness, self.saturation, self.hue = brightness, saturation, hue

    def _crop_params(self, w: int, h: int):
        area = w * h
        log_r = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        for _ in range(10):
            target = area * _uniform(*self.scale)
            ar = math.exp(_uniform(*log_r))
            cw, ch = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
            if 0 < cw <= w and 0 < ch <= h:
                top = int(torch.randint(0, h - ch + 1, (1,)).item())
                left = int(torch.randint(0, w - cw + 1, (1,)).item())
                return top, left, ch, cw
        in_ratio = w / h
        if in_ratio < self.ratio[0]:
            cw, ch = w, int(round(w / self.ratio[0]))
        elif in_ratio > self.ratio[1]:
            ch, cw = h, int(round(h * self.ratio[1]))
        else:
            cw, ch = w, h
        return (h - ch) // 2, (w - cw) // 2, ch, cw

    @staticmethod
    def _hue(img: Image.Image, delta: float) -> Image.Image:
        hsv = np.array(img.convert("HSV"), dtype=np.uint8)
        hsv[..., 0] = (hsv[..., 0].astype(np.int16) + int(delta * 255)) % 256
        return Image.fromarray(hsv, "HSV").convert("RGB")

    def __call__(self, image: Image.Image, boxes: List[List[float]]):
        w, h = image.size
        top, left, ch, cw = self._crop_params(w, h)
        image = image.crop((left, top, left + cw, top + ch)).resize((self.size[1], self.size[0]), Image.BILINEAR)
        sx, sy = self.size[1] / cw, self.size[0] / ch
        out = []
        for x0, y0, x1, y1 in boxes:
            x0, x1 = min(max(x0 - left, 0.0), cw) * sx, min(max(x1 - left, 0.0), cw) * sx
            y0, y1 = min(max(y0 - top, 0.0), ch) * sy, min(max(y1 - top, 0.0), ch) * sy
            out.append([x0, y0, x1, y1])
        ops = []
        if self.brightness:
            f = _uniform(max(0.0, 1 - self.brightness), 1 + self.brightness)
            ops.append(lambda im, f=f: ImageEnhance.Brightness(im).enhance(f))
        if self.saturation:
            f = _uniform(max(0.0, 1 - self.saturation), 1 + self.saturation)
            ops.append(lambda im, f=f: ImageEnhance.Color(im).enhance(f))
        if self.hue:
            f = _uniform(-self.hue, self.hue)
            ops.append(lambda im, f=f: self._hue(im, f))
        for k in torch.randperm(len(ops)).tolist():
            image = ops[k](image)
        return image, out


class VOCDetectionYOLO(Dataset):
    """PASCAL VOC detection samples as (image tensor (3, H, W), target (S, S, 5B+C)); reference dataset.py:16-588."""

    VOC_CLASSES = VOC_CLASSES
    split_paths = {
        "2007": {"trainval": "VOCtrainval_06-Nov-2007", "test": "VOCtest_06-Nov-2007", "train": "VOCtrainval_06-Nov-2007",
                 "val": "VOCtrainval_06-Nov-2007"},
        "2012": {"trainval": "VOCtrainval_11-May-2012", "test": "VOCtest_11-May-2012", "train": "VOCtrainval_11-May-2012",
                 "val": "VOCtrainval_11-May-2012"},
    }

    @staticmethod
    def download_from_kaggle(year: str = "2007", verbose: bool = True):
        raise ImportError("download_from_kaggle needs kagglehub and a network connection; place the dataset under `root` "
                          "(VOCdevkit/VOC<year>/...) and pass download=False")

    def __init__(self, root: str | Path = None, year: str = "2007", image_set: str = "train", download: bool = False, S: int = 7, B: int = 2,
                 transform=None, target_size: Tuple[int, int] = (448, 448), augment: bool = True):
        self.S, self.B = S, B
        self.C = len(self.VOC_CLASSES)
        self.target_size = target_size
        self.augment = augment and image_set == "train"            # only the training split is augmented (dataset.py:190)
        self.class_to_idx = {n: i for i, n in enumerate(self.VOC_CLASSES)}
        self.class_names = self.VOC_CLASSES
        if download:
            self.download_from_kaggle(year.split("-")[0])
        if root is None:
            raise FileNotFoundError("VOCDetectionYOLO needs `root` (the dataset cannot be downloaded offline)")
        base_year = year.split("-")[0]
        root = Path(root)
        cands = [root / self.split_paths[base_year][image_set] / "VOCdevkit" / f"VOC{base_year}", root / "VOCdevkit" / f"VOC{base_year}",
                 root / f"VOC{base_year}", root]
        self.voc_dir = next((c for c in cands if (c / "ImageSets" / "Main" / f"{image_set}.txt").is_file()), None)
        if self.voc_dir is None:
            raise FileNotFoundError(f"no ImageSets/Main/{image_set}.txt for VOC{base_year} under {root} (looked in: "
                                    + ", ".join(str(c) for c in cands) + ")")
        with open(self.voc_dir / "ImageSets" / "Main" / f"{image_set}.txt") as f:
            self.ids = [ln.split()[0] for ln in f if ln.strip()]
        if transform is not None:
            self.transform = transform
        elif self.augment:
            self.transform = self._get_augmentation_transforms()
        else:
            from .inference import _Preprocess
            self.transform = _Preprocess(size=target_size)
        from .inference import _Preprocess
        self._finish = _Preprocess(size=target_size)                # ToTensor + Normalize (the resize is a no-op after the crop)

    def _get_augmentation_transforms(self):
        return _Augment(self.target_size)

    def __len__(self) -> int:
        return len(self.ids)

    def _load(self, idx: int):
        name = self.ids[idx]
        image = Image.open(self.voc_dir / "JPEGImages" / f"{name}.jpg").convert("RGB")
        annotation = parse_voc_xml(ET.parse(self.voc_dir / "Annotations" / f"{name}.xml").getroot())
        return image, annotation

    def __getitem__(self, idx: int):
        image, annotation = self._load(idx)
        if self.augment and isinstance(self.transform, _Augment):
            bboxes, class_ids = self._extract_bboxes_from_annotation(annotation)
            w, h = image.size
            pix = [[(x - bw / 2) * w, (y - bh / 2) * h, (x + bw / 2) * w, (y + bh / 2) * h] for x, y, bw, bh in bboxes]
            image, pix = self.transform(image, pix)
            H, W = self.target_size
            norm = []
            for x0, y0, x1, y1 in pix:
                clamp = lambda v: max(0, min(1, v))   # noqa: E731
                norm.append([clamp(((x0 + x1) / 2) / W), clamp(((y0 + y1) / 2) / H), clamp((x1 - x0) / W), clamp((y1 - y0) / H)])
            return self._finish(image), self._encode_target(norm, class_ids)
        return self.transform(image), self._parse_voc_annotation(annotation)

    # ---- annotation handling: same names / arguments / results as the reference (dataset.py:411-532)
    def _extract_bboxes_from_annotation(self, annotation: dict):
        size = annotation["annotation"]["size"]
        iw, ih = float(size["width"]), float(size["height"])
        objects = annotation["annotation"].get("object", [])
        if not isinstance(objects, list):
            objects = [objects]
        bboxes, class_ids = [], []
        for obj in objects:
            if obj["name"] not in self.class_to_idx:
                continue
            bb = obj["bndbox"]
            xmin, ymin, xmax, ymax = float(bb["xmin"]), float(bb["ymin"]), float(bb["xmax"]), float(bb["ymax"])
            vals = [((xmin + xmax) / 2.0) / iw, ((ymin + ymax) / 2.0) / ih, (xmax - xmin) / iw, (ymax - ymin) / ih]
            bboxes.append([max(0, min(1, v)) for v in vals])
            class_ids.append(self.class_to_idx[obj["name"]])
        return bboxes, class_ids

    def _parse_voc_annotation(self, annotation: dict) -> torch.Tensor:
        return self._encode_target(*self._extract_bboxes_from_annotation(annotation))

    def _encode_target(self, bboxes: list, class_ids: list) -> torch.Tensor:
        return encode_target(bboxes, class_ids, self.S, self.B, self.C)

    def visualize_sample(self, idx: int) -> dict:
        image, annotation = self._load(idx)
        bboxes, class_ids = self._extract_bboxes_from_annotation(annotation)
        return {"image_path": str(self.voc_dir / "JPEGImages" / f"{self.ids[idx]}.jpg"), "image_size": image.size, "bboxes": bboxes,
                "class_ids": class_ids, "class_names": [self.class_names[c] for c in class_ids], "num_objects": len(bboxes)}


class CombinedVOCDataset(Dataset):
    """Concatenation of several VOCDetectionYOLO datasets with identical S / B / C (dataset.py:590-659)."""

    def __init__(self, datasets: list):
        self.datasets = datasets
        self.lengths = [len(ds) for ds in datasets]
        self.cumulative_lengths = np.cumsum([0] + self.lengths).tolist()
        if datasets:
            first = datasets[0]
            self.S, self.B, self.C = first.S, first.B, first.C
            self.class_names, self.class_to_idx = first.class_names, first.class_to_idx
            for ds in datasets[1:]:
                assert ds.S == self.S, f"All datasets must have same S (grid size): {self.S} != {ds.S}"
                assert ds.B == self.B, f"All datasets must have same B (boxes per cell): {self.B} != {ds.B}"
                assert ds.C == self.C, f"All datasets must have same C (num classes): {self.C} != {ds.C}"

    def __len__(self) -> int:
        return sum(self.lengths)

    def __getitem__(self, idx: int):
        if idx < 0 or idx >= len(self):
            raise IndexError(f"Index {idx} out of range for dataset of size {len(self)}")
        k = int(np.searchsorted(self.cumulative_lengths, idx, side="right")) - 1
        return self.datasets[k][idx - self.cumulative_lengths[k]]


def create_voc_datasets(years_and_splits: list, download: bool = True, S: int = 7, B: int = 2, target_size: Tuple[int, int] = (448, 448),
                        augment: bool = True, root: str | Path = None) -> Dataset:
    """One VOCDetectionYOLO, or their concatenation, for [(year, image_set), ...] (dataset.py:662-730).  Offline, ``download``
    is honoured only as "the data must already lie under root" (default root: $VOC_ROOT or ./data)."""
    if root is None:
        root = os.environ.get("VOC_ROOT", "data")
    datasets = [VOCDetectionYOLO(root=root, year=y, image_set=s, download=False, S=S, B=B, target_size=target_size, augment=augment)
                for y, s in years_and_splits]
    return datasets[0] if len(datasets) == 1 else CombinedVOCDataset(datasets)

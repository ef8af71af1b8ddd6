"""TextualInversionDataset with the sample contract of the reference's training/dataset.py:605-739:

    {pixel_values (3,H,W) f32 in [-1,1], input_ids (77,) i64, input_ids_placeholder_object,
     input_ids_placeholder_view, text, image_idx}

Covered: learnable_mode 0 (object only; any image folder) and the DTU `dtu-12d` view modes 1-5
(view tokens generated from the calibration matrices exactly like dataset.py:411-514: the token string
carries the 12 camera numbers rounded to 4 decimals with '.' -> 'p').  The augmentation pipelines
(`augmentation_key` 1-8, dataset.py:238-316) come from `augment.py` (PIL/numpy restatement: torchvision
is absent from this image).
Resizing uses PIL bicubic like the reference (the Coach never overrides `interpolation`).
"""
from __future__ import annotations

import random
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch
from PIL import Image

from .constants import DTU_SPLIT_IDXS, IMAGENET_TEMPLATES_SMALL, PATH_DTU_CALIBRATION_DIR
from .utils_utils import filter_paths_imgs, num_to_string, string_to_num


class TextualInversionDataset(torch.utils.data.Dataset):
    def __init__(self, data_root: Path, tokenizer, camera_representation: str = "spherical", learnable_mode: int = 0,
                 train_data_subsets: List[Path] = None, placeholder_object_tokens: List[str] = None,
                 fixed_object_token_or_path=None, size: int = 512, repeats: int = 100, flip_p: float = 0.0,
                 set: str = "train", placeholder_object_token: str = "*", dtu_lighting: str = "3", dtu_subset: int = 0,
                 caption_strategy: int = 0, dtu_preprocess_key: int = 0, augmentation_key: int = 0,
                 center_crop: bool = False, device_pipeline: bool = False):
        self.learnable_mode = learnable_mode
        self.device_pipeline = device_pipeline  # images are produced on the GPU from the plan this dataset draws
        self.data_root = Path(data_root)
        self.tokenizer = tokenizer
        self.size = size
        self.placeholder_object_token = placeholder_object_token
        self.center_crop = center_crop
        self.flip_p = flip_p if learnable_mode == 0 else 0  # dataset.py:95-96
        self.train_data_subsets = train_data_subsets
        self.camera_representation = camera_representation
        self.dtu_lighting = str(dtu_lighting)
        self.dtu_subset = dtu_subset
        self.dtu_preprocess_key = dtu_preprocess_key
        self.caption_strategy = caption_strategy
        self.augmentation_key = augmentation_key
        self.augmentations = None
        if augmentation_key > 0:
            from .augment import build_augmentations
            if learnable_mode == 0:
                aug_size = (size, size)
            elif dtu_preprocess_key == 0:
                aug_size = (512, 512)
            elif dtu_preprocess_key == 1:
                aug_size = (384, 512)  # (height, width): reversed w.r.t. PIL's size (dataset.py:236)
            else:
                raise NotImplementedError("augmentation with dtu_preprocess_key 2 is undefined in the reference "
                                          "(dataset.py:231-236 leaves `size` unset)")
            self.augmentations = build_augmentations(augmentation_key, aug_size)
            self.aug_size = aug_size
        if learnable_mode != 3:
            paths = filter_paths_imgs(sorted(self.data_root.glob("*")))
            if camera_representation == "dtu-12d" and learnable_mode != 0:
                paths = self.dtu_filter_fnames_lighting(paths, self.dtu_lighting)
                paths = self.dtu_filter_image_paths_from_idx(paths, self.dtu_get_train_idxs(dtu_subset))
            self.image_paths = paths
            self.image_paths_flattened = paths
        else:
            self.image_paths = {}
            for sub in train_data_subsets:
                p = filter_paths_imgs(sorted((self.data_root / str(sub)).glob("*")))
                if camera_representation == "dtu-12d":
                    p = self.dtu_filter_fnames_lighting(p, self.dtu_lighting)
                    p = self.dtu_filter_image_paths_from_idx(p, self.dtu_get_train_idxs(dtu_subset))
                assert len(p) > 0
                self.image_paths[str(sub)] = p
            self.image_paths_flattened = [q for row in self.image_paths.values() for q in row]
            self.current_object_idx = np.random.choice(len(train_data_subsets))
        self.num_images = len(self.image_paths_flattened)
        assert self.num_images > 0, "no .png/.jpg images found. Check the --data.train_data_dir option"
        self._length = self.num_images * repeats if set == "train" else self.num_images
        self.templates = IMAGENET_TEMPLATES_SMALL
        if learnable_mode == 0:
            self.placeholder_object_tokens = [placeholder_object_token]
            self.placeholder_view_tokens: List[str] = []
            self.fixed_object_token = None
        else:
            if camera_representation != "dtu-12d":
                raise NotImplementedError("view modes are implemented for camera_representation='dtu-12d'")
            self.lookup_camidx_to_view_token, self.lookup_camidx_to_cam_params = self.dtu_generate_dset_cam_tokens_params()
            self.lookup_view_token_to_camidx = {v: k for k, v in self.lookup_camidx_to_view_token.items()}
            cams = np.unique([self.dtu_cam_info_from_fname(f)[0] for f in self.image_paths_flattened])
            self.placeholder_view_tokens = [self.lookup_camidx_to_view_token[k] for k in sorted(cams)]
            self.fixed_object_token = fixed_object_token_or_path if learnable_mode == 1 else None
            if learnable_mode == 1:
                self.placeholder_object_tokens = []
            elif learnable_mode == 3:
                self.placeholder_object_tokens = placeholder_object_tokens
                self.lookup_object_to_placeholder_object_token = {str(s): t for s, t in
                                                                  zip(train_data_subsets, placeholder_object_tokens)}
            else:
                self.placeholder_object_tokens = [placeholder_object_token]
        self.placeholder_tokens = self.placeholder_view_tokens + self.placeholder_object_tokens

    # ------------------------------------------------------------------ DTU helpers (dataset.py:320-522)
    @staticmethod
    def dtu_get_train_idxs(dtu_subset):
        tr = DTU_SPLIT_IDXS["train"]
        table = {0: tr + DTU_SPLIT_IDXS["test"], 1: tr[:1], 3: tr[:3], 6: tr[:6], 9: tr,
                 -1: list(range(12, 36)), -2: list(range(12, 36, 2)), -3: list(range(12, 36, 3))}
        if dtu_subset not in table:
            raise NotImplementedError()
        return table[dtu_subset]

    @staticmethod
    def dtu_filter_fnames_lighting(image_paths, dtu_lighting):
        return [f for f in image_paths if Path(f).stem.split("_")[2] == str(dtu_lighting)]

    @staticmethod
    def dtu_cam_info_from_fname(fname):
        cam, light = Path(fname).stem.split("_")[1:3]
        return int(cam) - 1, light  # file names are 1-indexed, camera keys 0-indexed

    @staticmethod
    def dtu_cam_and_lighting_to_fname(cam_idx, lighting_idx):
        return f"rect_{cam_idx + 1:03d}_{lighting_idx}_r5000.png"

    @staticmethod
    def dtu_filter_image_paths_from_idx(image_paths, idxs):
        key = lambda f: TextualInversionDataset.dtu_cam_info_from_fname(f)[0]
        return sorted([f for f in image_paths if key(f) in idxs], key=key)

    @staticmethod
    def dtu_cam_params_to_token(cam_params: torch.Tensor, cam_key="NULL") -> str:
        p = cam_params.flatten()
        assert len(p) == 12
        return f"<view_dtu12d_cam{cam_key}_" + "_".join(num_to_string(n.item(), tol=4) for n in p) + ">"

    @staticmethod
    def dtu_token_to_cam_params(view_token: str, cam_idx_as_int: bool = False):
        cam_idx = view_token.split("_")[2][3:]
        if cam_idx_as_int:
            cam_idx = int(cam_idx)
        return torch.tensor([string_to_num(n) for n in view_token[:-1].split("_")[3:]]), cam_idx

    @staticmethod
    def read_text_file_to_tensor(file_path):
        with open(file_path) as f:
            return torch.tensor([[float(x) for x in line.split()] for line in f if line.strip()])

    @staticmethod
    def dtu_generate_dset_cam_tokens_params(calibration_dir: str = PATH_DTU_CALIBRATION_DIR):
        tok, par = {}, {}
        for f in Path(calibration_dir).iterdir():
            if f.suffix != ".txt":
                continue
            key = int(f.stem.split("_")[1]) - 1
            assert key not in par
            par[key] = TextualInversionDataset.read_text_file_to_tensor(f)
            tok[key] = TextualInversionDataset.dtu_cam_params_to_token(par[key], key)
        return tok, par

    def reset_sampled_object(self):
        """mode 3: draw the scene every micro-batch from the global numpy stream (coach.py:155-156,
        dataset.py:584-600)."""
        assert self.learnable_mode == 3
        self.current_object_idx = np.random.choice(len(self.train_data_subsets))

    # ------------------------------------------------------------------ samples
    def __len__(self) -> int:
        return self._length

    def _resize(self, image: Image.Image) -> Image.Image:
        if "dtu" in str(self.data_root):
            if self.dtu_preprocess_key == 0:
                canvas = Image.new("RGB", (image.size[0], image.size[1] + 400), "black")
                canvas.paste(image, (0, 0))
                return canvas.resize((512, 512), resample=Image.BICUBIC)
            if self.dtu_preprocess_key == 1:
                return image.resize((512, 384), resample=Image.BICUBIC)
            if self.dtu_preprocess_key == 2:
                return image.resize((768, 576), resample=Image.BICUBIC)
            raise NotImplementedError()
        if "llff" in str(self.data_root):
            return image
        return image.resize((self.size, self.size), resample=Image.BICUBIC)

    def _source_array(self, image: Image.Image) -> np.ndarray:
        arr = np.array(image).astype(np.uint8)
        if self.center_crop:
            h, w = arr.shape[:2]
            c = min(h, w)
            arr = arr[(h - c) // 2:(h + c) // 2, (w - c) // 2:(w + c) // 2]
        return arr

    def _source_hw(self, hw):
        if self.center_crop:
            c = min(hw)
            return (c, c)
        return tuple(hw)

    def target_size(self):
        """(height, width) `_resize` produces, None when it leaves the image alone (llff)"""
        if "dtu" in str(self.data_root):
            return {0: (512, 512), 1: (384, 512), 2: (576, 768)}[self.dtu_preprocess_key]
        if "llff" in str(self.data_root):
            return None
        return (self.size, self.size)

    def load_source(self, path: str) -> np.ndarray:
        """the uint8 image `_resize` starts from (RGB, centre crop, the 400 black rows of dtu_preprocess_key 0):
        what the device pipeline caches in HBM"""
        image = Image.open(path)
        if image.mode != "RGB":
            image = image.convert("RGB")
        arr = self._source_array(image)
        if "dtu" in str(self.data_root) and self.dtu_preprocess_key == 0:
            arr = np.concatenate([arr, np.zeros((400, arr.shape[1], 3), np.uint8)], 0)
        return arr

    @staticmethod
    def collate(samples):
        """default collation, except that the augmentation plans stay python objects"""
        from torch.utils.data import default_collate
        aug = [s.pop("aug") for s in samples] if "aug" in samples[0] else None
        batch = default_collate(samples)
        if aug is not None:
            batch["aug"] = aug
        return batch

    def __getitem__(self, i: int) -> Dict[str, Any]:
        if self.learnable_mode != 3:
            paths = self.image_paths
            obj_token = self.placeholder_object_tokens[0] if self.placeholder_object_tokens else None
            idx = i % self.num_images
        else:
            cur = str(self.train_data_subsets[self.current_object_idx])
            paths = self.image_paths[cur]
            obj_token = self.lookup_object_to_placeholder_object_token[cur]
            idx = i % len(paths)
        path = paths[idx]
        image = Image.open(path)  # lazy: decoded only on the host path below
        if image.mode != "RGB" and not self.device_pipeline:
            image = image.convert("RGB")
        ex: Dict[str, Any] = {"image_idx": idx}
        template = random.choice(self.templates)  # drawn even when unused, like the reference (:630)
        if self.learnable_mode == 0:
            ex["text"] = template.format(obj_token)
            ex["input_ids_placeholder_view"] = torch.tensor(-1)
            ex["input_ids_placeholder_object"] = torch.tensor(self.tokenizer.convert_tokens_to_ids(obj_token))
        else:
            cam_key, _ = self.dtu_cam_info_from_fname(path)
            view_token = self.lookup_camidx_to_view_token[cam_key]
            assert view_token in self.placeholder_view_tokens
            if self.learnable_mode == 1:
                ex["text"] = f"{view_token}. A photo of a {self.fixed_object_token}"
                ex["input_ids_placeholder_object"] = torch.tensor(-1)
            else:
                ex["text"] = f"{view_token}. A photo of a {obj_token}"
                ex["input_ids_placeholder_object"] = torch.tensor(self.tokenizer.convert_tokens_to_ids(obj_token))
            ex["input_ids_placeholder_view"] = torch.tensor(self.tokenizer.convert_tokens_to_ids(view_token))
        ex["input_ids"] = self.tokenizer(ex["text"], padding="max_length", truncation=True,
                                         max_length=self.tokenizer.model_max_length, return_tensors="pt").input_ids[0]
        if self.device_pipeline:
            # same random draws in the same order as the host path below; the pixels are produced by
            # engine/input_pipeline.py in the training process from the cached source image
            flip = bool(self.learnable_mode == 0 and self.flip_p > 0 and torch.rand(1).item() < self.flip_p)
            plan = []
            if self.augmentations is not None:
                from .augment import draw_plan
                tgt = self.target_size() or self._source_hw(image.size[::-1])
                plan = draw_plan(self.augmentation_key, self.aug_size, tgt[1], tgt[0])
            ex["aug"] = dict(path=str(path), flip=flip, plan=plan)
            return ex
        image = self._resize(Image.fromarray(self._source_array(image)))
        if self.learnable_mode == 0 and self.flip_p > 0 and torch.rand(1).item() < self.flip_p:
            image = image.transpose(Image.FLIP_LEFT_RIGHT)
        if self.augmentations is not None:
            img_size = image.size
            image = self.augmentations(image)
            assert image.size == img_size  # dataset.py:731-732
        arr = (np.array(image).astype(np.uint8) / 127.5 - 1.0).astype(np.float32)
        ex["pixel_values"] = torch.from_numpy(arr).permute(2, 0, 1)
        return ex

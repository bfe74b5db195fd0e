"""Training / rendering ray source on device-resident dataset tensors.

Mirrors the ray-facing part of the reference's ``ScenePixelSource`` (datasets/base/pixel_source.py): ``get_rays``
(:39-76), ``sample_uniform_rays`` (:622-668), ``sample_important_rays`` (:564-620), ``get_train_rays`` (:670-731),
``get_render_rays`` (:733-826, full resolution), ``build_pixel_error_buffer`` / ``update_pixel_error_maps`` (:462-517).
Loading images / poses from disk (the dataset classes proper) stays out of scope: the constructor takes the tensors the
reference's loaders produce.

Every per-batch operation is a HIP kernel (csrc/rays.hip): one launch for uniform pixels, a radix-select race for the
error-buffer multinomial (no host round trip), one gather kernel for rays + colours + masks + timestamps.  Random numbers
come from a counter-based generator keyed by a seed word in device memory (``self.seed_word``), advanced by one tiny
device op per batch, so a captured hipGraph replays with fresh rays.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from .ops import _check_cuda, _ptr, _stream

_SALT_UNIFORM, _SALT_RACE, _SALT_CELL = 0x1111, 0x2222, 0x3333


def get_rays(x: Tensor, y: Tensor, c2w: Tensor, intrinsic: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """pixel_source.py:39-76 for per-ray camera matrices: x, y [n] (any integer / float dtype), c2w [n,4,4] or [4,4],
    intrinsic [n,3,3] or [3,3] -> (origins [n,3], viewdirs [n,3], direction_norm [n,1])."""
    _check_cuda(x, y, c2w, intrinsic)
    n = x.numel()
    dev = x.device
    c2w = c2w.reshape(-1, 4, 4).float().contiguous()
    K = intrinsic.reshape(-1, 3, 3).float().contiguous()
    if c2w.shape[0] == 1 and K.shape[0] == 1:
        idx = torch.zeros(n, device=dev, dtype=torch.int64)
    else:
        assert c2w.shape[0] == n and K.shape[0] == n, "one camera per ray (or a single camera for all rays)"
        idx = torch.arange(n, device=dev, dtype=torch.int64)
    xi, yi = x.reshape(-1).to(torch.int64).contiguous(), y.reshape(-1).to(torch.int64).contiguous()
    o, d = torch.empty((n, 3), device=dev), torch.empty((n, 3), device=dev)
    nrm = torch.empty((n, 1), device=dev)
    with torch.cuda.device(dev):
        _lib.call("emer_gen_rays", _ptr(idx), _ptr(yi), _ptr(xi), _ptr(c2w), _ptr(K), None, None, None, None, n, 1, 1, _ptr(o), _ptr(d),
                  _ptr(nrm), None, None, None, None, None, _stream(o))
    return o, d, nrm


class PixelSource:
    """images [n_imgs,H,W,3] fp32 in [0,1], cam_to_worlds [n_imgs,4,4], intrinsics [n_imgs,3,3] (all on the GPU);
    optional sky_masks [n_imgs,H,W], normalized_timestamps [n_imgs], cam_ids [n_imgs]."""

    def __init__(self, images: Tensor, cam_to_worlds: Tensor, intrinsics: Tensor, sky_masks: Optional[Tensor] = None,
                 normalized_timestamps: Optional[Tensor] = None, cam_ids: Optional[Tensor] = None, buffer_ratio: float = 0.0,
                 buffer_downscale: int = 4, seed: int = 0) -> None:
        _check_cuda(images, cam_to_worlds, intrinsics, sky_masks, normalized_timestamps, cam_ids)
        self.images = images.float().contiguous()
        self.num_imgs, self.HEIGHT, self.WIDTH = self.images.shape[:3]
        self.cam_to_worlds = cam_to_worlds.float().contiguous()
        self.intrinsics = intrinsics.float().contiguous()
        self.sky_masks = None if sky_masks is None else sky_masks.float().contiguous()
        self.normalized_timestamps = None if normalized_timestamps is None else normalized_timestamps.float().contiguous()
        self.cam_ids = None if cam_ids is None else cam_ids.to(torch.int64).contiguous()
        self.device = self.images.device
        self.buffer_ratio, self.buffer_downscale = float(buffer_ratio), int(buffer_downscale)
        self.pixel_error_maps: Optional[Tensor] = None
        self.pixel_error_buffered = False
        self.seed_word = torch.tensor([seed], dtype=torch.int64, device=self.device)  # read by the kernels as uint64
        self._ws = torch.empty(4 + 2048, dtype=torch.int32, device=self.device)
        self._all = None

    # ------------------------------------------------------------------------------------ error buffer (:462-517)
    def build_pixel_error_buffer(self) -> None:
        if self.buffer_ratio > 0:
            self.pixel_error_maps = torch.ones((self.num_imgs, self.HEIGHT // self.buffer_downscale, self.WIDTH // self.buffer_downscale),
                                               dtype=torch.float32, device=self.device)

    def update_pixel_error_maps(self, pred_rgbs: Tensor, gt_rgbs: Tensor, dynamic_opacities: Optional[Tensor] = None) -> None:
        """|gt - pred| averaged over colour at the buffer's resolution, x5 where the dynamic opacity exceeds 0.1,
        normalised to [0, 1] (offline: runs once per evaluation, plain torch)."""
        if self.pixel_error_maps is None:
            return
        err = (gt_rgbs.to(self.device) - pred_rgbs.to(self.device)).abs().mean(dim=-1)
        assert err.shape == self.pixel_error_maps.shape
        if dynamic_opacities is not None:
            err = torch.where(dynamic_opacities.to(self.device).reshape(err.shape) > 0.1, err * 5, err)
        self.pixel_error_maps = ((err - err.min()) / (err.max() - err.min())).contiguous()
        self.pixel_error_buffered = True

    # ------------------------------------------------------------------------------------------------ sampling
    def _candidates(self, cand) -> Tuple[Optional[Tensor], int]:
        if cand is None:
            return None, self.num_imgs
        if not isinstance(cand, Tensor):
            cand = torch.tensor(cand)
        cand = cand.to(self.device, torch.int64).contiguous()
        return cand, cand.numel()

    def _next_seed(self) -> None:
        self.seed_word.add_(0x9E3779B9)  # device-side: graph-capturable

    def sample_uniform_rays(self, num_rays: int, img_candidate_indices=None, advance: bool = True) -> Tuple[Tensor, Tensor, Tensor]:
        """:622-668 -> (img_id, y, x), int64 [num_rays].  Every call draws fresh pixels (the seed word advances on the device;
        ``advance=False``: the caller advances it once for a group of draws, as get_train_rays does)."""
        cand, n_c = self._candidates(img_candidate_indices)
        img, y, x = (torch.empty(num_rays, dtype=torch.int64, device=self.device) for _ in range(3))
        with torch.cuda.device(self.device):
            _lib.call("emer_sample_uniform", _ptr(self.seed_word), _SALT_UNIFORM, num_rays, _ptr(cand), n_c, self.HEIGHT, self.WIDTH,
                      _ptr(img), _ptr(y), _ptr(x), _stream(img))
        if advance:
            self._next_seed()
        return img, y, x

    def sample_important_rays(self, num_rays: int, img_candidate_indices=None, advance: bool = True,
                              check_support: bool = True) -> Tuple[Tensor, Tensor, Tensor]:
        """:564-620: multinomial over the error buffer of the candidate images WITHOUT replacement, then a random pixel of
        the chosen buffer cell.  Like torch.multinomial, asking for more samples than there are cells of positive weight is
        an error (``check_support``: one device->host read; a captured graph passes False after checking once)."""
        assert self.pixel_error_buffered, "Pixel error buffer not built."
        cand, n_c = self._candidates(img_candidate_indices)
        maps = self.pixel_error_maps if cand is None else self.pixel_error_maps[cand].contiguous()
        if check_support:
            n_pos = int((maps > 0).sum())
            if num_rays > n_pos:
                raise RuntimeError(f"sample_important_rays: cannot draw {num_rays} cells without replacement from {n_pos} cells of positive weight")
        Hb, Wb = maps.shape[1:]
        flat = torch.empty(num_rays, dtype=torch.int64, device=self.device)
        img, y, x = (torch.empty(num_rays, dtype=torch.int64, device=self.device) for _ in range(3))
        with torch.cuda.device(self.device):
            st = _stream(flat)
            _lib.call("emer_sample_importance", _ptr(maps), maps.numel(), _ptr(self.seed_word), _SALT_RACE, num_rays, _ptr(self._ws), _ptr(flat), st)
            _lib.call("emer_buffer_to_pixels", _ptr(flat), num_rays, Hb, Wb, self.buffer_downscale, _ptr(cand), self.HEIGHT, self.WIDTH,
                      _ptr(self.seed_word), _SALT_CELL, _ptr(img), _ptr(y), _ptr(x), st)
        if advance:
            self._next_seed()
        return img, y, x

    def _support_ok(self, num_rays: int, img_candidate_indices=None) -> bool:
        """True when the cached count of positive cells of the WHOLE error buffer already proves that ``num_rays`` cells can be
        drawn without replacement (candidate subsets fall back to the per-call check)."""
        if img_candidate_indices is not None:
            return False
        ver = self.pixel_error_maps._version
        cache = getattr(self, "_support_cache", None)
        if cache is None or cache[0] != ver or cache[2] is not self.pixel_error_maps:
            cache = self._support_cache = (ver, int((self.pixel_error_maps > 0).sum()), self.pixel_error_maps)
        return num_rays <= cache[1]

    # ---------------------------------------------------------------------------------------------------- rays
    def _gather(self, img_idx: Tensor, y: Tensor, x: Tensor) -> Dict[str, Tensor]:
        n, dev = img_idx.numel(), self.device
        out = {"origins": torch.empty((n, 3), device=dev), "viewdirs": torch.empty((n, 3), device=dev),
               "direction_norms": torch.empty((n, 1), device=dev), "pixel_coords": torch.empty((n, 2), device=dev),
               "pixels": torch.empty((n, 3), device=dev)}
        sky = torch.empty(n, device=dev) if self.sky_masks is not None else None
        ts = torch.empty(n, device=dev) if self.normalized_timestamps is not None else None
        cam = torch.empty(n, dtype=torch.int64, device=dev) if self.cam_ids is not None else None
        with torch.cuda.device(dev):
            _lib.call("emer_gen_rays", _ptr(img_idx), _ptr(y), _ptr(x), _ptr(self.cam_to_worlds), _ptr(self.intrinsics), _ptr(self.images),
                      _ptr(self.sky_masks), _ptr(self.normalized_timestamps), _ptr(self.cam_ids), n, self.HEIGHT, self.WIDTH,
                      _ptr(out["origins"]), _ptr(out["viewdirs"]), _ptr(out["direction_norms"]), _ptr(out["pixel_coords"]),
                      _ptr(out["pixels"]), _ptr(sky), _ptr(ts), _ptr(cam), _stream(img_idx))
        if ts is not None:
            out["normed_timestamps"] = ts
        out["img_idx"] = img_idx
        if cam is not None:
            out["cam_idx"] = cam
        if sky is not None:
            out["sky_masks"] = sky
        return out

    def get_train_rays(self, num_rays: int, candidate_indices=None) -> Dict[str, Tensor]:
        """:670-731: ``buffer_ratio`` of the batch from the error buffer (once it exists), the rest uniform."""
        if self.buffer_ratio > 0 and self.pixel_error_buffered:
            n_roi = int(num_rays * self.buffer_ratio)
            ri, ry, rx = self.sample_uniform_rays(num_rays - n_roi, candidate_indices, advance=False)
            # support of the error buffer: one device->host read when the buffer is built / updated (cached), none per step
            bi, by, bx = self.sample_important_rays(n_roi, candidate_indices, advance=False,
                                                    check_support=not self._support_ok(n_roi, candidate_indices))
            img_idx, y, x = torch.cat([ri, bi]), torch.cat([ry, by]), torch.cat([rx, bx])
        else:
            img_idx, y, x = self.sample_uniform_rays(num_rays, candidate_indices, advance=False)
        self._next_seed()
        return self._gather(img_idx, y, x)

    def get_render_rays(self, img_idx: int) -> Dict[str, Tensor]:
        """:733-826 at full resolution: every pixel of one image, image-shaped tensors (H, W, ...)."""
        H, W, dev = self.HEIGHT, self.WIDTH, self.device
        if self._all is None:
            yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
            self._all = (yy.reshape(-1).contiguous(), xx.reshape(-1).contiguous())
        y, x = self._all
        out = self._gather(torch.full((H * W,), int(img_idx), dtype=torch.int64, device=dev), y, x)
        out["direction_norm"] = out.pop("direction_norms")   # the reference's render dict spells this key in the singular (:836)
        return {k: v.reshape(H, W, -1).squeeze(-1) if k in ("normed_timestamps", "img_idx", "cam_idx", "sky_masks") else v.reshape(H, W, -1)
                for k, v in out.items()}

    def __len__(self) -> int:
        return self.num_imgs

    def __getitem__(self, idx: int) -> Dict[str, Tensor]:
        return self.get_render_rays(idx)

    # -------------------------------------------------------------------------------------------- synthetic data
    @classmethod
    def synthetic(cls, device, num_imgs: int = 50, height: int = 160, width: int = 240, num_cams: int = 1, seed: int = 0,
                  buffer_ratio: float = 0.0) -> "PixelSource":
        """A seeded stand-in for a driving log (no dataset on the box): a camera moving along +x through the scene box of
        configs/default_config.yaml, random images, 15 % sky."""
        g = torch.Generator().manual_seed(seed)
        n_t = num_imgs // num_cams
        c2w = torch.eye(4).repeat(num_imgs, 1, 1)
        for i in range(num_imgs):
            t, cam = i // num_cams, i % num_cams
            yaw = (cam - (num_cams - 1) / 2) * 0.7
            # camera looks along world +x (OpenCV camera: z forward, x right, y down)
            fwd = torch.tensor([torch.cos(torch.tensor(yaw)), torch.sin(torch.tensor(yaw)), 0.0])
            right = torch.tensor([fwd[1], -fwd[0], 0.0])
            down = torch.tensor([0.0, 0.0, -1.0])
            c2w[i, :3, 0], c2w[i, :3, 1], c2w[i, :3, 2] = right, down, fwd
            c2w[i, :3, 3] = torch.tensor([60.0 * t / max(n_t - 1, 1), 0.0, 2.0])
        K = torch.tensor([[0.8 * width, 0.0, width / 2], [0.0, 0.8 * width, height / 2], [0.0, 0.0, 1.0]]).repeat(num_imgs, 1, 1)
        images = torch.rand(num_imgs, height, width, 3, generator=g)
        sky = (torch.rand(num_imgs, height, width, generator=g) < 0.15).float()
        ts = (torch.arange(num_imgs) // num_cams).float() / max(n_t - 1, 1)
        cams = torch.arange(num_imgs) % num_cams
        return cls(images.to(device), c2w.to(device), K.to(device), sky.to(device), ts.to(device), cams.to(device),
                   buffer_ratio=buffer_ratio, seed=seed)

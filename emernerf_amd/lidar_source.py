"""Lidar training / rendering ray source on device-resident point tensors (second half).

Mirrors the ray-facing part of the reference's ``SceneLidarSource`` (datasets/base/lidar_source.py): the timestamp registry
(:139-221), ``sample_uniform_rays`` with its cached per-timestep subset (:223-275), ``get_train_rays`` (:277-308) and
``get_render_rays`` (:310-330).  Loading scans and calibrations from disk (the dataset classes proper, datasets/waymo.py:229-336)
stays out of scope: the constructor takes the tensors the reference's loaders produce -- origins / unit directions [n, 3], ranges
[n, 1] (``torch.norm(..., keepdim=True)``, waymo.py:293), integer timesteps [n].

The per-batch work -- ``torch.randint`` over the cached scans and the four gathers of ``get_train_rays``, five launches in the
reference -- is one HIP kernel (``emer_lidar_sample_rays``, csrc/rays.hip) on the counter-based generator of ``PixelSource``
(seed word in device memory, advanced by a device op, so a captured hipGraph replays with fresh rays).  The cached subset is
rebuilt only when the candidate timesteps change (plain torch indexing: once per split, not per step).

Reference behaviour kept on purpose:
  * ``get_train_rays`` ALWAYS gathers from the cached subset (:295-298).  With ``candidate_indices=None`` the reference draws indices
    over ALL points (:241-244) and then indexes the cache with them -- a TypeError before any cache exists and out-of-range reads
    after.  The split wrapper always passes its timesteps (datasets/base/split_wrapper.py:31), so that path is never taken; here it
    raises ``EmerError`` instead of reading out of range.
  * a candidate list given as a Tensor is compared with the cache directly; given as a list it is converted first (:246-262).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from . import _lib
from .ops import _check_cuda, _ptr, _stream

_SALT_LIDAR = 0x4444


class LidarSource:
    """origins [n,3], directions [n,3] (unit), ranges [n,1] or [n], timesteps [n] int64 -- all on the GPU; optional
    normalized_timestamps [n] in [0, 1] (else ``register_normalized_timestamps`` later, as the reference's dataset does)."""

    def __init__(self, origins: Tensor, directions: Tensor, ranges: Tensor, timesteps: Tensor,
                 normalized_timestamps: Optional[Tensor] = None, seed: int = 0) -> None:
        _check_cuda(origins, directions, ranges, timesteps, normalized_timestamps)
        n = origins.shape[0]
        assert origins.shape == (n, 3) and directions.shape == (n, 3) and ranges.numel() == n and timesteps.numel() == n
        self.device = origins.device
        self.origins = origins.float().contiguous()
        self.directions = directions.float().contiguous()
        self.ranges = ranges.float().contiguous()          # shape kept: the gathers return (num_rays, *ranges.shape[1:])
        self._timesteps = timesteps.reshape(-1).to(torch.int64).contiguous()
        self._normalized_timestamps: Optional[Tensor] = None
        self._unique_normalized_timestamps: Optional[Tensor] = None
        if normalized_timestamps is not None:
            self.register_normalized_timestamps(normalized_timestamps)
        self.cached_indices: Optional[Tensor] = None
        self.cached_origins = self.cached_directions = self.cached_ranges = self.cached_normalized_timestamps = None
        self.seed_word = torch.tensor([seed], dtype=torch.int64, device=self.device)   # read by the kernel as uint64

    # ------------------------------------------------------------------------------------- timestamp registry (:139-221)
    @property
    def timesteps(self) -> Tensor:
        return self._timesteps

    @property
    def normalized_timestamps(self) -> Tensor:
        return self._normalized_timestamps

    @property
    def unique_normalized_timestamps(self) -> Tensor:
        return self._unique_normalized_timestamps

    @property
    def num_timesteps(self) -> int:
        return len(self.timesteps.unique())

    def register_normalized_timestamps(self, normalized_timestamps: Tensor) -> None:
        assert normalized_timestamps.size(0) == self.origins.size(0), \
            "The number of lidar points and the number of normalized timestamps must match."
        assert normalized_timestamps.min() >= 0 and normalized_timestamps.max() <= 1, "The normalized timestamps must be in the range [0, 1]."
        self._normalized_timestamps = normalized_timestamps.to(self.device).float().contiguous()
        self._unique_normalized_timestamps = self._normalized_timestamps.unique()

    def find_closest_timestep(self, normed_timestamp: float) -> Tensor:
        return torch.argmin(torch.abs(self.unique_normalized_timestamps - normed_timestamp))

    # ----------------------------------------------------------------------------------------------- sampling (:223-275)
    def _rebuild_cache(self, candidate_indices: Tensor) -> None:
        self.cached_indices = candidate_indices
        mask = torch.isin(self.timesteps, candidate_indices.to(self.timesteps.dtype))   # the reference ORs one comparison per index
        self.cached_origins = self.origins[mask].contiguous()
        self.cached_directions = self.directions[mask].contiguous()
        self.cached_ranges = self.ranges[mask].contiguous()
        self.cached_normalized_timestamps = None if self._normalized_timestamps is None else self._normalized_timestamps[mask].contiguous()

    def _update_cache(self, candidate_indices) -> None:
        if not isinstance(candidate_indices, Tensor):
            candidate_indices = torch.tensor(candidate_indices, device=self.device)
            if self.cached_indices is None:
                self._rebuild_cache(candidate_indices)
        if self.cached_indices is None:
            # (the reference reaches torch.equal(tensor, None) here and raises a TypeError)
            raise TypeError("candidate_indices given as a Tensor before any cache exists: pass a list first (lidar_source.py:246-263)")
        candidate_indices = candidate_indices.to(self.device)
        if not torch.equal(candidate_indices, self.cached_indices):
            self._rebuild_cache(candidate_indices)

    def _draw(self, num_rays: int, n_points: int, gather_from=None, idx_in: Optional[Tensor] = None):
        """One launch: the indices (drawn, or ``idx_in``) and the rays they select from ``gather_from`` = (origins, directions, ranges,
        timestamps) (default: all points)."""
        dev = self.device
        idx = torch.empty(num_rays, dtype=torch.int64, device=dev)
        o, d, r, t = gather_from if gather_from is not None else (self.origins, self.directions, self.ranges, self._normalized_timestamps)
        oo, od = torch.empty((num_rays, 3), device=dev), torch.empty((num_rays, 3), device=dev)
        orr = torch.empty((num_rays,) + tuple(self.ranges.shape[1:]), device=dev)
        ot = torch.empty(num_rays, device=dev) if t is not None else None
        with torch.cuda.device(dev):
            _lib.call("emer_lidar_sample_rays", _ptr(self.seed_word), _SALT_LIDAR, num_rays, int(n_points), _ptr(idx_in), _ptr(o), _ptr(d),
                      _ptr(r), _ptr(t), _ptr(idx), _ptr(oo), _ptr(od), _ptr(orr), _ptr(ot), _stream(idx))
        return idx, oo, od, orr, ot

    def _next_seed(self) -> None:
        self.seed_word.add_(0x9E3779B9)   # device-side: graph-capturable

    def sample_uniform_rays(self, num_rays: int, candidate_indices=None) -> Tensor:
        """:223-275 -> indices [num_rays] int64: over all points without candidates, else over the cached subset of the candidate
        timesteps (rebuilt when they change)."""
        if candidate_indices is None:
            idx = self._draw(num_rays, self.origins.shape[0])[0]
        else:
            self._update_cache(candidate_indices)
            idx = self._draw(num_rays, self.cached_origins.shape[0],
                             (self.cached_origins, self.cached_directions, self.cached_ranges, self.cached_normalized_timestamps))[0]
        self._next_seed()
        return idx

    # --------------------------------------------------------------------------------------------------- rays (:277-330)
    def get_train_rays(self, num_rays: int, candidate_indices=None, lidar_idx: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """:277-308: a batch of ``num_rays`` rays of the candidate scans, keys ``lidar_origins / lidar_viewdirs / lidar_ranges /
        lidar_normed_timestamps``.  ``lidar_idx`` (not in the reference's signature): gather these indices of the cached subset instead
        of drawing -- a recording of the reference's own draw is replayed through it (tests/test_golden_gpu.py)."""
        if candidate_indices is not None:
            self._update_cache(candidate_indices)
        if self.cached_origins is None:
            raise _lib.EmerError("get_train_rays gathers from the cached scans (lidar_source.py:295-298): pass candidate_indices (the "
                                 "split's timesteps) at least once")
        assert self.cached_normalized_timestamps is not None, "register_normalized_timestamps first"
        if lidar_idx is not None:
            lidar_idx = lidar_idx.to(self.device, torch.int64).contiguous()
            assert int(lidar_idx.max()) < self.cached_origins.shape[0] and int(lidar_idx.min()) >= 0
            num_rays = lidar_idx.numel()
        _, o, d, r, t = self._draw(num_rays, self.cached_origins.shape[0],
                                   (self.cached_origins, self.cached_directions, self.cached_ranges, self.cached_normalized_timestamps),
                                   idx_in=lidar_idx)
        if lidar_idx is None:
            self._next_seed()
        return {"lidar_origins": o, "lidar_viewdirs": d, "lidar_ranges": r, "lidar_normed_timestamps": t}

    def get_render_rays(self, time_idx: int) -> Dict[str, Tensor]:
        """:310-330: every point of scan ``time_idx`` (boolean-mask selection: once per rendered frame, plain torch)."""
        sel = self.timesteps == time_idx
        return {"lidar_origins": self.origins[sel], "lidar_viewdirs": self.directions[sel], "lidar_ranges": self.ranges[sel],
                "lidar_normed_timestamps": self.normalized_timestamps[sel]}

    # -------------------------------------------------------------------------------------------- synthetic data
    @classmethod
    def synthetic(cls, device, num_timesteps: int = 50, points_per_scan: int = 4096, seed: int = 0) -> "LidarSource":
        """A seeded stand-in for a log's lidar sweeps (no dataset on the box): a spinning sensor 2 m above the ego path along +x
        through the scene box of configs/default_config.yaml, ranges ~U(2, 70) m."""
        g = torch.Generator().manual_seed(seed)
        n = num_timesteps * points_per_scan
        ts = torch.arange(num_timesteps).repeat_interleave(points_per_scan)
        o = torch.stack([60.0 * ts.float() / max(num_timesteps - 1, 1), torch.zeros(n), torch.full((n,), 2.0)], -1)
        az, el = torch.rand(n, generator=g) * 6.2831853, (torch.rand(n, generator=g) - 0.7) * 0.45
        d = torch.stack([torch.cos(az) * torch.cos(el), torch.sin(az) * torch.cos(el), torch.sin(el)], -1)
        rng = (torch.rand(n, generator=g) * 68 + 2)[:, None]
        nts = ts.float() / max(num_timesteps - 1, 1)
        return cls(o.to(device), d.to(device), rng.to(device), ts.to(device), nts.to(device), seed=seed)

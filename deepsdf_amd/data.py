"""SDF sample files -> training batches.

On-disk format (the reference's, deep_sdf/data.py:74-80 and sdf_sampler/sdf_sampler.py:146): one ``.npz`` per scene with arrays
``pos`` / ``neg`` of shape [*, G+1] (xyz..., sdf), float32 or float64, NaN rows possible.

``DeviceSampleCache`` is the MI355X-side replacement of the reference's ``SDFSamples`` dataset + DataLoader workers
(deep_sdf/data.py:142-194): every scene's filtered samples are uploaded ONCE into HBM (288 GB holds thousands
of 50k-point scenes) and each step's balanced subsample is drawn on the device, so no per-step np.load / H2D copy remains.
There is deliberately no host-side ``Dataset`` here: the trainer never iterates samples on the CPU.  What stays on the host
is the file list (``get_instance_filenames``), the reader of one scene file (``load_scene``) and -- for tools that want one
draw without a GPU, and for pinning the count rule to the reference (golden G12) -- ``unpack_sdf_samples``.
"""
import logging
import os

import numpy as np
import torch

from . import workspace as ws


def get_instance_filenames(data_source, split):
    """Scene files of a split, in split order; a missing file only warns (deep_sdf/data.py:15-33)."""
    npzfiles = []
    for dataset in split:
        for class_name in split[dataset]:
            for instance_name in split[dataset][class_name]:
                instance_filename = os.path.join(dataset, class_name, instance_name + ".npz")
                if not os.path.isfile(os.path.join(data_source, ws.sdf_samples_subdir, instance_filename)):
                    logging.warning("Requested non-existent file '{}'".format(instance_filename))
                npzfiles += [instance_filename]
    return npzfiles


def remove_nans(tensor, geom_dimension):
    """Rows whose sdf column is NaN are dropped; the rest as fp32 (deep_sdf/data.py:61-63)."""
    return tensor[~torch.isnan(tensor[:, geom_dimension])].float()


def load_scene(path, geom_dimension):
    """One scene file -> (positives, negatives): fp32 [n, G+1] tensors without NaN rows."""
    with np.load(path) as npz:
        return tuple(remove_nans(torch.from_numpy(npz[k]), geom_dimension) for k in ("pos", "neg"))


def _balanced_counts(n_pos, n_neg, subsample):
    """Rows per sign of one draw: half each, a shortfall of one sign is made up by the other (deep_sdf/data.py:83-91; pinned to
    the reference's loader by golden G12)."""
    half = int(subsample / 2)
    if n_pos < half:
        return n_pos, 2 * half - n_pos
    if n_neg < half:
        return 2 * half - n_neg, n_neg
    return half, half


def unpack_sdf_samples(filename, geom_dimension, subsample=None, generator=None):
    """One draw from one scene file ON THE HOST: positives first, then negatives, each without replacement; everything when
    subsample is None (deep_sdf/data.py:74-110).  The trainer does not use this (DeviceSampleCache.sample draws on the GPU)."""
    pos, neg = load_scene(filename, geom_dimension)
    if subsample is None:
        return torch.cat([pos, neg], 0)
    n_pos, n_neg = _balanced_counts(len(pos), len(neg), subsample)
    pick = lambda t, n: t[torch.randperm(len(t), generator=generator)[:n]]   # noqa: E731  (fewer rows than n: all of them, as the reference)
    return torch.cat([pick(pos, n_pos), pick(neg, n_neg)], 0)


class DeviceSampleCache:
    """All scenes' samples resident in HBM; balanced without-replacement subsampling on the device in ONE kernel launch
    (``dsdf_sample_batch``, include/dsdf.h): the per-step replacement of the reference's DataLoader workers running
    ``unpack_sdf_samples`` (deep_sdf/data.py:74-110) + collate (train_deep_sdf.py:483-501).

    Layout: one [total_rows, G+1] fp32 tensor; scene k's positives are rows [pos_start[k], pos_start[k]+n_pos[k]),
    negatives likewise.  ``sample(scene_ids, S)`` returns xyz [B*S', G], sdf [B*S'] with S' = 2*(S//2): per scene S'/2
    positives then S'/2 negatives (a shortfall of one sign is taken from the other), each drawn without replacement by
    the keyed permutation ``oracle.sample_perm`` specifies.  GPU only: there is no CPU path.
    """

    def __init__(self, tensors_pos_neg, geom_dimension, device):
        self.G = geom_dimension
        self.device = torch.device(device)
        rows, self.n_pos, self.n_neg, self.pos_start, self.neg_start = [], [], [], [], []
        off = 0
        for pos, neg in tensors_pos_neg:
            self.pos_start.append(off); self.n_pos.append(pos.shape[0]); off += pos.shape[0]
            self.neg_start.append(off); self.n_neg.append(neg.shape[0]); off += neg.shape[0]
            rows += [pos[:, :geom_dimension + 1], neg[:, :geom_dimension + 1]]
        if max(max(self.n_pos), max(self.n_neg)) > (1 << 30):
            raise ValueError("more than 2^30 samples of one sign in a scene")
        self.data = torch.cat(rows, 0).to(self.device, torch.float32).contiguous()
        as_dev = lambda x: torch.tensor(x, dtype=torch.int64, device=self.device)  # noqa: E731
        self.n_pos_d, self.n_neg_d = as_dev(self.n_pos), as_dev(self.n_neg)
        self.pos_start_d, self.neg_start_d = as_dev(self.pos_start), as_dev(self.neg_start)
        self._draws = 0

    @staticmethod
    def from_files(data_source, npzfiles, geom_dimension, device):
        items = [load_scene(os.path.join(data_source, ws.sdf_samples_subdir, f), geom_dimension) for f in npzfiles]
        return DeviceSampleCache(items, geom_dimension, device)

    def __len__(self):
        return len(self.n_pos)

    def draw_key(self, generator=None):
        """64-bit key of the next draw: the generator's seed (host value, no device sync) mixed with a draw counter."""
        seed = generator.initial_seed() if generator is not None else 0
        self._draws += 1
        return ((seed * self.KEY_SEED_MUL) + self._draws * self.KEY_DRAW_MUL) & ((1 << 64) - 1)

    KEY_SEED_MUL, KEY_DRAW_MUL = 0x9E3779B97F4A7C15, 0xD1B54A32D192ED03

    def _check_ids(self, scene_ids, S):
        for k in scene_ids.tolist():
            if not 0 <= k < len(self.n_pos):
                raise IndexError(f"scene {k} is not in the cache")
            if self.n_pos[k] + self.n_neg[k] < S:
                raise ValueError(f"scene {k} has {self.n_pos[k] + self.n_neg[k]} samples, fewer than the {S} requested")

    def sample_sequence(self, scene_ids, subsample, n_draws, counter, xyz_out, sdf_out, generator=None):
        """The draws of a graph-captured loop: returns launch(), which enqueues ONE draw whose key is that of draw number `*counter`
        (a device int64 the loop advances) of the next `n_draws` draws of this cache -- the keys, and therefore the batches, that
        n_draws calls of sample() with the same generator would have produced.  xyz_out [B*S', G] / sdf_out [B*S'] are
        overwritten by every launch (dsdf_sample_batch_seq)."""
        if self.device.type != "cuda":
            raise RuntimeError("DeviceSampleCache.sample_sequence runs on the GPU; there is no CPU path")
        from . import _lib
        from .engine import _ptr, _stream
        scene_ids = torch.as_tensor(scene_ids, dtype=torch.int64)
        S = 2 * int(subsample / 2)
        self._check_ids(scene_ids, S)
        B = scene_ids.numel()
        sid = scene_ids.to(self.device)
        seed = generator.initial_seed() if generator is not None else 0
        key0 = (seed * self.KEY_SEED_MUL + (self._draws + 1) * self.KEY_DRAW_MUL) & ((1 << 64) - 1)
        self._draws += int(n_draws)

        def launch():
            _lib.check(_lib.lib().dsdf_sample_batch_seq(_ptr(self.data), self.G, _ptr(self.pos_start_d), _ptr(self.n_pos_d),
                                                        _ptr(self.neg_start_d), _ptr(self.n_neg_d), _ptr(sid), B, int(subsample), key0,
                                                        self.KEY_DRAW_MUL, _ptr(counter), _ptr(xyz_out), _ptr(sdf_out), _stream()))
        return launch

    def sample(self, scene_ids, subsample, generator=None, key=None, scene_ids_device=None):
        """scene_ids: [B] integer tensor (CPU preferred: its values are checked on the host) -> (xyz [B*S', G], sdf [B*S']).
        scene_ids_device: the same ids already on the GPU (int64) -- saves the per-call host->device copy, which on pageable
        memory makes the host wait for all queued GPU work."""
        if self.device.type != "cuda":
            raise RuntimeError("DeviceSampleCache.sample runs on the GPU (libdsdf_hip.so dsdf_sample_batch); there is no CPU path")
        from . import _lib
        from .engine import _ptr, _stream
        scene_ids = torch.as_tensor(scene_ids, dtype=torch.int64)
        S = 2 * int(subsample / 2)
        self._check_ids(scene_ids, S)
        B = scene_ids.numel()
        sid = scene_ids.to(self.device) if scene_ids_device is None else scene_ids_device
        xyz = torch.empty(B * S, self.G, dtype=torch.float32, device=self.device)
        sdf = torch.empty(B * S, dtype=torch.float32, device=self.device)
        key = self.draw_key(generator) if key is None else int(key) & ((1 << 64) - 1)
        _lib.check(_lib.lib().dsdf_sample_batch(_ptr(self.data), self.G, _ptr(self.pos_start_d), _ptr(self.n_pos_d),
                                                _ptr(self.neg_start_d), _ptr(self.n_neg_d), _ptr(sid), B, int(subsample), key,
                                                _ptr(xyz), _ptr(sdf), _stream()))
        return xyz, sdf


class StagedSampleCache:
    """DeviceSampleCache for datasets LARGER than the HBM budget (SURVEY 8 f1 beyond the resident cache; the reference's
    `load_ram` mode of deep_sdf/data.py:142-194 is its nearest relative): every scene's filtered samples live in PINNED host
    memory, and only the scenes of the batches in flight are on the device.  sample() uploads its batch's scenes on a COPY
    stream into one of `depth` device windows and launches the same sampling kernel on it; the trainer draws batch i + 1 before it
    runs step i (deepsdf_amd/train.py), so the upload of the next batch overlaps the current step's compute.

    Same interface and -- for the same generator / key -- the SAME BATCHES as DeviceSampleCache: the kernel still indexes by the
    global scene id (its permutation keys depend on it); per window a full-length pos_start / neg_start array holds window-relative
    row offsets, of which only the batch's entries are rewritten per draw.  A window is reused only after the sampling launch that
    read it has finished (events), so `depth` draws may be in flight.  GPU only."""

    def __init__(self, tensors_pos_neg, geom_dimension, device, max_batch_scenes, depth=2):
        self.G = geom_dimension
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("StagedSampleCache needs a CUDA/HIP device (pinned host memory + copy stream); there is no CPU path")
        rows, self.n_pos, self.n_neg, self.host_start = [], [], [], []
        off = 0
        for pos, neg in tensors_pos_neg:
            self.host_start.append(off)
            self.n_pos.append(pos.shape[0]); self.n_neg.append(neg.shape[0])
            off += pos.shape[0] + neg.shape[0]
            rows += [pos[:, :geom_dimension + 1], neg[:, :geom_dimension + 1]]
        if max(max(self.n_pos), max(self.n_neg)) > (1 << 30):
            raise ValueError("more than 2^30 samples of one sign in a scene")
        self.host = torch.cat(rows, 0).to(torch.float32).contiguous().pin_memory()          # [total_rows, G+1], scene after scene
        n = len(self.n_pos)
        as_dev = lambda x: torch.tensor(x, dtype=torch.int64, device=self.device)  # noqa: E731
        self.n_pos_d, self.n_neg_d = as_dev(self.n_pos), as_dev(self.n_neg)
        self.max_batch = int(max_batch_scenes)
        per_scene = sorted((a + b for a, b in zip(self.n_pos, self.n_neg)), reverse=True)
        cap = sum(per_scene[:self.max_batch])                                                 # rows of the largest possible batch
        self.depth = int(depth)
        self.windows = [torch.empty(cap, geom_dimension + 1, dtype=torch.float32, device=self.device) for _ in range(self.depth)]
        self.pos_start_w = [torch.zeros(n, dtype=torch.int64, device=self.device) for _ in range(self.depth)]
        self.neg_start_w = [torch.zeros(n, dtype=torch.int64, device=self.device) for _ in range(self.depth)]
        self.stage = [torch.empty(3, self.max_batch, dtype=torch.int64).pin_memory() for _ in range(self.depth)]   # ids | pos_start | neg_start
        self.stage_d = [torch.empty(3, self.max_batch, dtype=torch.int64, device=self.device) for _ in range(self.depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.ready = [torch.cuda.Event() for _ in range(self.depth)]
        self.free = [None] * self.depth            # event after the sampling launch that last read the window
        self.stage_free = [None] * self.depth      # event after the copy that last read the pinned staging rows
        self._turn = 0
        self._draws = 0
        self.uploaded_bytes = 0

    @staticmethod
    def from_files(data_source, npzfiles, geom_dimension, device, max_batch_scenes, depth=2):
        items = [load_scene(os.path.join(data_source, ws.sdf_samples_subdir, f), geom_dimension) for f in npzfiles]
        return StagedSampleCache(items, geom_dimension, device, max_batch_scenes, depth)

    def __len__(self):
        return len(self.n_pos)

    draw_key = DeviceSampleCache.draw_key
    KEY_SEED_MUL, KEY_DRAW_MUL = DeviceSampleCache.KEY_SEED_MUL, DeviceSampleCache.KEY_DRAW_MUL

    def sample(self, scene_ids, subsample, generator=None, key=None, scene_ids_device=None):
        """As DeviceSampleCache.sample (scene_ids_device is accepted and ignored: the ids travel with the offsets)."""
        from . import _lib
        from .engine import _ptr, _stream
        scene_ids = torch.as_tensor(scene_ids, dtype=torch.int64).cpu()
        S = 2 * int(subsample / 2)
        ids = scene_ids.tolist()
        B = len(ids)
        if B > self.max_batch:
            raise ValueError(f"batch of {B} scenes, the staging windows were sized for {self.max_batch}")
        for k in ids:
            if not 0 <= k < len(self.n_pos):
                raise IndexError(f"scene {k} is not in the cache")
            if self.n_pos[k] + self.n_neg[k] < S:
                raise ValueError(f"scene {k} has {self.n_pos[k] + self.n_neg[k]} samples, fewer than the {S} requested")
        w = self._turn % self.depth
        self._turn += 1
        compute = torch.cuda.current_stream(self.device)
        win, st, st_d = self.windows[w], self.stage[w], self.stage_d[w]
        if self.stage_free[w] is not None:
            self.stage_free[w].synchronize()       # the host rewrites pinned staging rows: their last upload must be done
        off = 0
        seen = {}
        for b, k in enumerate(ids):                # window-relative offsets (a scene drawn twice in a batch is uploaded once)
            if k not in seen:
                seen[k] = off
                off += self.n_pos[k] + self.n_neg[k]
            st[0, b], st[1, b], st[2, b] = k, seen[k], seen[k] + self.n_pos[k]
        with torch.cuda.stream(self.copy_stream):
            if self.free[w] is not None:
                self.copy_stream.wait_event(self.free[w])          # the launch that read this window has finished
            for k, o in seen.items():
                n = self.n_pos[k] + self.n_neg[k]
                win[o:o + n].copy_(self.host[self.host_start[k]:self.host_start[k] + n], non_blocking=True)
                self.uploaded_bytes += n * (self.G + 1) * 4
            st_d[:, :B].copy_(st[:, :B], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
            self.stage_free[w] = ev
            self.pos_start_w[w].index_copy_(0, st_d[0, :B], st_d[1, :B])
            self.neg_start_w[w].index_copy_(0, st_d[0, :B], st_d[2, :B])
            self.ready[w].record(self.copy_stream)
        compute.wait_event(self.ready[w])
        xyz = torch.empty(B * S, self.G, dtype=torch.float32, device=self.device)
        sdf = torch.empty(B * S, dtype=torch.float32, device=self.device)
        key = self.draw_key(generator) if key is None else int(key) & ((1 << 64) - 1)
        _lib.check(_lib.lib().dsdf_sample_batch(_ptr(win), self.G, _ptr(self.pos_start_w[w]), _ptr(self.n_pos_d),
                                                _ptr(self.neg_start_w[w]), _ptr(self.n_neg_d), _ptr(st_d[0]), B, int(subsample), key,
                                                _ptr(xyz), _ptr(sdf), _stream()))
        fe = torch.cuda.Event()
        fe.record(compute)
        self.free[w] = fe
        return xyz, sdf


def make_sample_cache(data_source, npzfiles, geom_dimension, device, max_batch_scenes):
    """The trainer's sample cache: resident in HBM (DeviceSampleCache) when the scenes fit the budget, staged through pinned host
    memory (StagedSampleCache) otherwise.  Budget: DSDF_SAMPLE_CACHE_GB (GiB; 0 forces staging), default 60 % of the HBM that is
    free when the trainer starts."""
    items = [load_scene(os.path.join(data_source, ws.sdf_samples_subdir, f), geom_dimension) for f in npzfiles]
    need = sum((p.shape[0] + n.shape[0]) * (geom_dimension + 1) * 4 for p, n in items)
    env = os.environ.get("DSDF_SAMPLE_CACHE_GB")
    if env is not None:
        budget = float(env) * (1 << 30)
    else:
        budget = 0.6 * torch.cuda.mem_get_info(torch.device(device))[0]
    if need <= budget:
        return DeviceSampleCache(items, geom_dimension, device)
    logging.info("sample cache: {:.1f} GiB of samples exceed the {:.1f} GiB HBM budget -> staged through pinned host memory".format(
        need / (1 << 30), budget / (1 << 30)))
    return StagedSampleCache(items, geom_dimension, device, max_batch_scenes)

"""Data feed of the pretraining step: the nnU-Net v2 preprocessed-folder reader and 3-D patch sampler the
reference drives through `nnUNetDataset` / `nnUNetDataLoader3D` (nnunetv2/training/dataloading/nnunet_dataset.py:11-146,
data_loader_3d.py:6-49, base_data_loader.py:10-139), restated without batchgenerators (absent here and on the GPU box).

Kept: the folder layout (`<case>.npy` [+ `<case>_seg.npy`] or `<case>.npz` with 'data'/'seg', `<case>.pkl` with
'class_locations'), batches `{'data','seg','properties','keys'}`, cases drawn with replacement, foreground oversampling of the
LAST round(B*(1-p)).. samples of a batch (p = 0.33, P/pretrain_AntoMask.py:337), the bounding-box rules with `need_to_pad`
(enlarged initial patch, compute_initial_patch_size.py:4-24), data padded with 0 and seg with -1 -- pinned bit for bit against the
reference's own loader (tests/golden/make_loader_fixtures.py imports it with a stub of the absent batchgenerators base class).
The train transforms the drivers enable (SpatialTransform rotation / scaling, MirrorTransform; the intensity transforms are
commented out in the reference, P/pretrain_AntoMask.py:99-109) run ON THE DEVICE (DeviceAugmenter -> csrc/aug_ops.hip); the host
side only draws their parameters.  PrefetchLoader + DeviceFeed keep the GPU fed (pinned, double-buffered H2D on a copy stream).
"""
import os
import pickle
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


class PreprocessedDataset:
    """keys() / load_case(key) over an nnU-Net v2 preprocessed folder (nnunet_dataset.py:11-146, read-only)."""

    def __init__(self, folder: str, case_identifiers: Optional[List[str]] = None):
        if case_identifiers is None:
            case_identifiers = sorted({f[:-4] for f in os.listdir(folder) if f.endswith(".pkl")})
        self.folder, self.cases = folder, list(case_identifiers)
        if not self.cases:
            raise FileNotFoundError(f"no <case>.pkl files in {folder}")

    def keys(self) -> List[str]:
        return list(self.cases)

    def __len__(self):
        return len(self.cases)

    def unpack(self, overwrite: bool = False) -> int:
        """nnunetv2.training.dataloading.utils.unpack_dataset: write `<case>.npy` / `<case>_seg.npy` next to every `<case>.npz` ONCE, so
        that load_case memory-maps the volume instead of inflating the whole archive for every sampled patch.  Returns the number of
        cases unpacked.  (Atomic per file: written under a temporary name, then renamed.)"""
        n = 0
        for key in self.cases:
            base = os.path.join(self.folder, key)
            if not os.path.isfile(base + ".npz") or (os.path.isfile(base + ".npy") and not overwrite):
                continue
            z = np.load(base + ".npz")
            for name, arr in (("", z["data"]), ("_seg", z["seg"] if "seg" in z.files else None)):
                if arr is None:
                    continue
                tmp = f"{base}{name}.tmp.npy"
                np.save(tmp, arr)
                os.replace(tmp, f"{base}{name}.npy")
            n += 1
        return n

    def load_case(self, key: str):
        base = os.path.join(self.folder, key)
        with open(base + ".pkl", "rb") as f:
            props = pickle.load(f)
        if os.path.isfile(base + ".npy"):                      # unpacked form (memory-mapped, as nnU-Net does after unpacking)
            data = np.load(base + ".npy", mmap_mode="r")
            seg = np.load(base + "_seg.npy", mmap_mode="r") if os.path.isfile(base + "_seg.npy") else None
        else:
            z = np.load(base + ".npz")
            data, seg = z["data"], (z["seg"] if "seg" in z.files else None)
        if seg is None:
            seg = np.full((1, *data.shape[1:]), -1, dtype=np.int16)
        return data, seg, props


class PinnedPool:
    """A fixed set of pinned host batch buffers shared by the loader threads: a loader writes its crops straight into one (no
    pageable -> pinned copy on the consumer's thread: that serial 14 ms memcpy per 4 x 205^3 batch capped the feed at 96 volumes/s),
    DeviceFeed hands it back once its host -> device copy has completed."""

    def __init__(self, shape, n: int, dtype=torch.float32):
        import queue
        self.free = queue.Queue()
        for _ in range(n):
            t = torch.empty(shape, dtype=dtype)
            self.free.put(t.pin_memory() if torch.cuda.is_available() else t)

    def get(self) -> torch.Tensor:
        return self.free.get()

    def put(self, t: torch.Tensor):
        self.free.put(t)


class PatchLoader3D:
    """nnUNetDataLoader3D.generate_train_batch (nnunetv2/training/dataloading/data_loader_3d.py:6-49) over get_bbox
    (base_data_loader.py:64-139), restated without batchgenerators: an infinite iterator of
    {'data' (B,C,*patch_size) fp32, 'seg' int16, 'properties', 'keys'} batches.

    `patch_size` is what is CROPPED (the reference passes the enlarged `initial_patch_size` of get_patch_size, :312-318),
    `final_patch_size` what the network gets after the spatial transform; need_to_pad = patch_size - final_patch_size lets the
    crop hang over the volume border (base_data_loader.py:28-35).  One RandomState drives exactly the reference's draws in the
    reference's order -- np.random.choice of the cases (batchgenerators DataLoader.get_indices, infinite=True), then per sample
    either randint per axis or choice(class) + choice(voxel) -- so a loader seeded like np.random.seed reproduces the reference's
    batches bit for bit (tests/golden/loader_tiny.npz).  Augmentation draws live in their own stream (SpatialAugmenter)."""

    def __init__(self, dataset: PreprocessedDataset, batch_size: int, patch_size: Sequence[int],
                 oversample_foreground_percent: float = 0.33, seed: int = 0, final_patch_size: Optional[Sequence[int]] = None,
                 pin_memory: bool = False, pool: Optional[PinnedPool] = None):
        self.ds, self.B, self.patch = dataset, batch_size, tuple(int(v) for v in patch_size)
        self.final = tuple(int(v) for v in (final_patch_size if final_patch_size is not None else patch_size))
        self.need_to_pad = [self.patch[d] - self.final[d] for d in range(3)]
        self.p_fg, self.rs, self.pin, self.pool = oversample_foreground_percent, np.random.RandomState(seed), pin_memory, pool
        import threading
        self.lock = threading.Lock()          # held while a batch draws from self.rs: rng_snapshot() sees the generator at a batch boundary
        self.keys = dataset.keys()

    def _force_fg(self, j: int) -> bool:                       # base_data_loader.py:47-51
        return not j < round(self.B * (1 - self.p_fg))

    def _bbox(self, shape, force_fg: bool, class_locations: Optional[Dict]):   # base_data_loader.py:64-139 (no ignore label)
        dim = len(shape)
        pad = list(self.need_to_pad)
        for d in range(dim):
            if pad[d] + shape[d] < self.patch[d]:
                pad[d] = self.patch[d] - shape[d]
        lbs = [-pad[d] // 2 for d in range(dim)]
        ubs = [shape[d] + pad[d] // 2 + pad[d] % 2 - self.patch[d] for d in range(dim)]
        voxel = None
        if force_fg:
            eligible = [k for k, v in class_locations.items() if len(v) > 0]
            if eligible:
                locs = class_locations[eligible[self.rs.choice(len(eligible))]]
                if len(locs) > 0:
                    voxel = locs[self.rs.choice(len(locs))]
        if voxel is not None:
            lb = [max(lbs[d], int(voxel[d + 1]) - self.patch[d] // 2) for d in range(dim)]
        else:
            lb = [int(self.rs.randint(lbs[d], ubs[d] + 1)) for d in range(dim)]
        return lb, [lb[d] + self.patch[d] for d in range(dim)]

    def __iter__(self):
        return self

    def rng_snapshot(self):
        """np.random.RandomState.get_state() of this loader, never in the middle of a batch's draws (the prefetch thread that owns the
        loader holds `lock` while it assembles a batch: key array and position are captured together)."""
        with self.lock:
            return self.rs.get_state()

    def __next__(self):
        with self.lock:
            return self._next_locked()

    def _next_locked(self):
        data_all = seg_all = None
        held = None
        props, sel = [], [self.keys[i] for i in self.rs.choice(len(self.keys), self.B, replace=True)]
        for j, key in enumerate(sel):
            data, seg, pr = self.ds.load_case(key)
            props.append(pr)
            if data_all is None:
                if self.pool is not None:                             # crops go straight into a pinned buffer of the shared pool
                    held = self.pool.get()
                    data_all = held.numpy()
                    data_all[...] = 0
                else:
                    data_all = np.zeros((self.B, data.shape[0], *self.patch), dtype=np.float32)
                seg_all = np.full((self.B, seg.shape[0], *self.patch), -1, dtype=np.int16)
            shape = data.shape[1:]
            lb, ub = self._bbox(shape, self._force_fg(j), pr.get("class_locations"))
            vlb = [max(0, lb[d]) for d in range(3)]
            vub = [min(shape[d], ub[d]) for d in range(3)]
            dst = tuple(slice(vlb[d] - lb[d], vlb[d] - lb[d] + (vub[d] - vlb[d])) for d in range(3))
            src = tuple(slice(vlb[d], vub[d]) for d in range(3))
            data_all[(j, slice(None)) + dst] = data[(slice(None),) + src]
            seg_all[(j, slice(None)) + dst] = seg[(slice(None),) + src]
        if self.pool is not None:
            return {"data": held, "seg": torch.from_numpy(seg_all), "properties": props, "keys": sel, "_release": (lambda t=held: self.pool.put(t))}
        d = torch.from_numpy(data_all)
        if self.pin and torch.cuda.is_available():
            d = d.pin_memory()
        return {"data": d, "seg": torch.from_numpy(seg_all), "properties": props, "keys": sel}


# --------------------------------------------------------------------------- spatial augmentation
def _rot_x(a):
    return np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])


def _rot_y(a):
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])


def _rot_z(a):
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])


def get_patch_size(final_patch_size, rot_x, rot_y, rot_z, scale_range):
    """nnunetv2/training/data_augmentation/compute_initial_patch_size.py:4-24: the enlarged crop that still covers the final patch
    after the largest rotation about each axis and the strongest zoom-out.  (rotate_coords_3d is batchgenerators': coords @ Rx Ry Rz.)"""
    r = [min(np.pi / 2, max(np.abs(v)) if isinstance(v, (tuple, list)) else v) for v in (rot_x, rot_y, rot_z)]
    coords = np.array(final_patch_size, dtype=np.float64)
    final_shape = coords.copy()
    for m in (_rot_x(r[0]), _rot_y(r[1]), _rot_z(r[2])):
        final_shape = np.max(np.vstack((np.abs(coords @ m), final_shape)), 0)
    final_shape /= min(scale_range)
    return final_shape.astype(int)


ROTATION_FOR_DA = (-30. / 360 * 2. * np.pi, 30. / 360 * 2. * np.pi)      # P/pretrain_AntoMask.py:303-307


class SpatialAugmenter:
    """Draws the per-sample parameters of the reference's train transforms (P/pretrain_AntoMask.py:78-113: SpatialTransform with
    do_rotation p 0.2 (all three axes, +-30 deg), do_scale p 0.2 in (0.7, 1.4), no elastic deformation, order-3 data interpolation,
    constant border 0, random_crop False; then MirrorTransform on axes (0, 1, 2) with p 0.5 each) and turns them into one 3x4 affine
    per sample: output voxel index -> coordinate in the enlarged patch.  batchgenerators (>= 0.25, unpinned, absent here) is restated
    from its published behaviour: zero-centred coordinate mesh, coords <- R^T coords with R = Rx Ry Rz, coords *= scale (scale < 1
    with probability 1/2), + centre of the enlarged patch.  Its RNG stream cannot be reproduced (worker processes, unseeded), so only
    the ARITHMETIC of a given draw is pinned (tests: against scipy.ndimage.map_coordinates)."""

    def __init__(self, final_patch_size, seed: int = 0, p_rot: float = 0.2, p_scale: float = 0.2, scale=(0.7, 1.4),
                 angle=ROTATION_FOR_DA, mirror_axes=(0, 1, 2), order: int = 3):
        self.final = tuple(int(v) for v in final_patch_size)
        self.rs = np.random.RandomState(seed)
        self.p_rot, self.p_scale, self.scale, self.angle, self.mirror_axes, self.order = p_rot, p_scale, scale, angle, mirror_axes, order

    def draw(self) -> dict:
        rs, p = self.rs, {"angles": (0.0, 0.0, 0.0), "scale": 1.0, "modified": False, "mirror": [False, False, False]}
        if rs.uniform() < self.p_rot:
            p["angles"] = tuple(float(rs.uniform(*self.angle)) for _ in range(3))
            p["modified"] = True
        if rs.uniform() < self.p_scale:
            lo, hi = self.scale
            p["scale"] = float(rs.uniform(lo, 1)) if (rs.uniform() < 0.5 and lo < 1) else float(rs.uniform(max(lo, 1), hi))
            p["modified"] = True
        for ax in self.mirror_axes:
            p["mirror"][ax] = bool(rs.uniform() < 0.5)
        return p

    def affine(self, p: dict, in_shape) -> np.ndarray:
        """3x4: source coordinate = A (o, 1) for output index o of the final patch (mirroring folded in as o -> n-1-o)."""
        ax, ay, az = p["angles"]
        R = (_rot_x(ax) @ _rot_y(ay) @ _rot_z(az)).T * p["scale"]
        ctr_out = np.array([(n - 1) / 2.0 for n in self.final])
        ctr_in = np.array([n / 2.0 - 0.5 for n in in_shape])
        if not p["modified"]:                      # batchgenerators centre-crops instead of interpolating: integer offsets
            ctr_in = np.array([(in_shape[d] - self.final[d]) // 2 + ctr_out[d] for d in range(3)])
            R = np.eye(3)
        M = np.eye(3)
        t = np.zeros(3)
        for d in range(3):
            if p["mirror"][d]:
                M[d, d] = -1.0
                t[d] = self.final[d] - 1
        A = np.zeros((3, 4))
        A[:, :3] = R @ M
        A[:, 3] = R @ (t - ctr_out) + ctr_in
        return A


class DeviceAugmenter:
    """Applies SpatialAugmenter draws ON THE GPU (am_spline_prefilter + am_resample_affine): enlarged device batch (B,1,De,He,We)
    -> (B,1,D,H,W).  Unmodified samples are integer crops / flips of the raw volume (order 0); rotated / scaled ones are prefiltered in
    a scratch copy and interpolated with order 3 (or 1)."""

    def __init__(self, aug: SpatialAugmenter):
        self.aug = aug
        self._scratch = None

    def __call__(self, x_enl: torch.Tensor, params: Optional[List[dict]] = None) -> torch.Tensor:
        from . import ops
        B, C = x_enl.shape[:2]
        assert C == 1 and x_enl.dtype == torch.float32 and x_enl.is_cuda
        params = params if params is not None else [self.aug.draw() for _ in range(B)]
        out = torch.empty(B, 1, *self.aug.final, device=x_enl.device, dtype=torch.float32)
        for b in range(B):
            src = x_enl[b, 0]
            A = self.aug.affine(params[b], src.shape)
            if params[b]["modified"] and self.aug.order == 3:
                if self._scratch is None or self._scratch.shape != src.shape:
                    self._scratch = torch.empty_like(src)
                self._scratch.copy_(src)
                src = ops.spline_prefilter(self._scratch)
            ops.resample_affine(src.contiguous(), out[b, 0], A.reshape(-1), self.aug.order if params[b]["modified"] else 0)
        return out


def gpu_numa_cpus(device_index: int) -> Optional[List[int]]:
    """CPUs of the NUMA node the GPU `device_index` hangs off (sysfs: /sys/bus/pci/devices/<bus id>/numa_node ->
    /sys/devices/system/node/node<N>/cpulist), or None when the platform does not say (single-node hosts report -1).
    One process per GPU: a rank's loader threads and its pinned host buffers belong next to ITS GPU's PCIe root -- on an 8-GPU node the
    other socket's memory costs a hop over the inter-socket link for every H2D batch (P/pretrain_AnatoMask_DDP.py:260-295 runs its
    workers wherever the scheduler puts them)."""
    try:
        import torch
        bus = torch.cuda.get_device_properties(device_index).pci_bus_id if hasattr(torch.cuda.get_device_properties(device_index), "pci_bus_id") else None
        cands = []
        if isinstance(bus, str):
            cands.append(bus.lower())
        elif bus is not None:
            pr = torch.cuda.get_device_properties(device_index)
            cands.append(f"{getattr(pr, 'pci_domain_id', 0):04x}:{int(bus):02x}:{getattr(pr, 'pci_device_id', 0):02x}.0")
        for c in cands:
            f = f"/sys/bus/pci/devices/{c}/numa_node"
            if os.path.exists(f):
                node = int(open(f).read().strip())
                if node < 0:
                    return None
                return parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
    except Exception:
        return None
    return None


def parse_cpulist(txt: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out: List[int] = []
    for part in txt.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


class PrefetchLoader:
    """Background producers around a batch iterator factory (the role of LimitedLenWrapper / NonDetMultiThreadedAugmenter with
    num_cached = 6, P/pretrain_AntoMask.py:343-345): `n_workers` threads each own a loader (their own RandomState: the reference's
    workers are unseeded and unordered too) and push batches into one bounded queue.  numpy crops / copies / np.load release the
    GIL, so threads scale; batches arrive in completion order."""

    def __init__(self, make_loader, n_workers: int = 4, num_cached: int = 6, cpus: Optional[List[int]] = None):
        """cpus: CPU ids the worker threads are pinned to (gpu_numa_cpus of the rank's GPU), None = wherever the scheduler puts them."""
        import queue
        import threading
        self.q = queue.Queue(maxsize=num_cached)
        self._stop = threading.Event()
        self.cpus = list(cpus) if cpus else None
        self.threads = [threading.Thread(target=self._run, args=(make_loader, w), daemon=True) for w in range(n_workers)]
        for t in self.threads:
            t.start()

    def _run(self, make_loader, w):
        if self.cpus:
            try:
                os.sched_setaffinity(0, set(self.cpus))     # (pid 0 = the calling thread on Linux)
            except OSError:
                pass
        it = iter(make_loader(w))
        while not self._stop.is_set():
            try:
                b = next(it)
            except StopIteration:
                break
            while not self._stop.is_set():
                try:
                    self.q.put(b, timeout=0.1)
                    break
                except Exception:                   # queue.Full
                    continue

    def __iter__(self):
        return self

    def __next__(self):
        return self.q.get()

    def close(self):
        self._stop.set()


class DeviceFeed:
    """Host -> device hand-off of the batches (the reference's `inp.to(device, non_blocking=True)` on pinned batches,
    P/pretrain_AntoMask.py:343-345,390-392), double-buffered: batch k+1 is copied on a dedicated HIP stream into the other
    device slot while step k computes; `next(feed)` returns the device tensor of batch k after making the compute stream wait
    for its copy event.  Un-pinned host batches are first copied into a persistent pinned staging buffer of the slot."""

    def __init__(self, batches, device, key: str = "data", depth: int = 2):
        self.it, self.dev, self.key, self.depth = iter(batches), device, key, depth
        self.copy_stream = torch.cuda.Stream(device=device)
        self.dbuf, self.pin, self.ready = [None] * depth, [None] * depth, [None] * depth
        self._pending = []                          # (copy event, release callback) of pooled pinned batches in flight
        self.k = 0
        self._start(0)

    def _start(self, slot: int, after=None) -> bool:
        # hand finished pool buffers back BEFORE blocking on the loader queue, and bound the host's lead over the GPU: the step has
        # no host synchronisation, so without the bound every pooled pinned buffer could sit in `_pending` behind a queued copy
        # while the loader threads starve in pool.get() and this thread waits forever in q.get()
        self._reap(self.depth)
        try:
            b = next(self.it)
        except StopIteration:
            self.ready[slot] = None
            return False
        h = b[self.key] if isinstance(b, dict) else b
        if self.dbuf[slot] is None or self.dbuf[slot].shape != h.shape:
            self.dbuf[slot] = torch.empty(h.shape, dtype=h.dtype, device=self.dev)
        if not h.is_pinned():
            if self.ready[slot] is not None:
                self.ready[slot].synchronize()             # the previous copy out of this staging buffer has finished
            if self.pin[slot] is None or self.pin[slot].shape != h.shape:
                self.pin[slot] = torch.empty(h.shape, dtype=h.dtype).pin_memory()
            self.pin[slot].copy_(h)
            h = self.pin[slot]
        with torch.cuda.stream(self.copy_stream):
            if after is not None:
                self.copy_stream.wait_event(after)         # the step that last read this device slot has been enqueued before `after`
            self.dbuf[slot].copy_(h, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.ready[slot] = ev
        self._host_keepalive = h
        if isinstance(b, dict) and "_release" in b:
            self._pending.append((ev, b["_release"]))
        return True

    def _reap(self, keep: int):
        """release the pool buffers whose copies have completed; wait for the oldest copies while more than `keep` are in flight."""
        self._pending = [(e, r) for e, r in self._pending if not (e.query() and (r() or True))]
        while len(self._pending) > keep:
            e, r = self._pending.pop(0)
            e.synchronize()
            r()

    def __iter__(self):
        return self

    def __next__(self) -> torch.Tensor:
        slot = self.k % self.depth
        if self.ready[slot] is None:
            raise StopIteration
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(self.ready[slot])
        out = self.dbuf[slot]
        # start the next copy into the other slot: it may only overwrite that slot once every kernel enqueued so far (the steps
        # that read it) has run -> an event recorded on the compute stream now
        done = torch.cuda.Event()
        done.record(cur)
        self.k += 1
        self._start(self.k % self.depth, after=done)
        return out

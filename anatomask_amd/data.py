"""Data feed of the pretraining step: the nnU-Net v2 preprocessed-folder reader and 3-D patch sampler the
reference drives through `nnUNetDataset` / `nnUNetDataLoader3D` (nnunetv2/training/dataloading/nnunet_dataset.py:11-146,
data_loader_3d.py:6-49, base_data_loader.py:10-139), restated without batchgenerators (absent here and on the GPU box).

Kept: the folder layout (`<case>.npy` [+ `<case>_seg.npy`] or `<case>.npz` with 'data'/'seg', `<case>.pkl` with
'class_locations'), batches `{'data','seg','properties','keys'}`, cases drawn with replacement, foreground oversampling of the
LAST round(B*(1-p)).. samples of a batch (p = 0.33, P/pretrain_AntoMask.py:337), the bounding-box rules (random corner inside
[-pad//2, shape+pad//2+pad%2-patch], foreground: a random voxel of a random class centred, clamped at the low side), data padded
with 0 and seg with -1.  Added: mirroring (the reference enables `MirrorTransform` on all axes, `:112-113`).  NOT restated: the
`SpatialTransform` rotations / scalings (p = 0.2 each, `:90-97`) -- parity of this module is unpinned (no reference run possible).
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


class PatchLoader3D:
    """Infinite iterator of nnU-Net style batches for `AnatoMaskTrainer.step(batch['data'])`."""

    def __init__(self, dataset: PreprocessedDataset, batch_size: int, patch_size: Sequence[int],
                 oversample_foreground_percent: float = 0.33, seed: int = 0, mirror_axes: Tuple[int, ...] = (0, 1, 2),
                 pin_memory: bool = True):
        self.ds, self.B, self.patch = dataset, batch_size, tuple(int(v) for v in patch_size)
        self.p_fg, self.rs, self.mirror_axes, self.pin = oversample_foreground_percent, np.random.RandomState(seed), mirror_axes, pin_memory
        self.keys = dataset.keys()

    def _force_fg(self, j: int) -> bool:                       # base_data_loader.py:47-51
        return not j < round(self.B * (1 - self.p_fg))

    def _bbox(self, shape, force_fg: bool, class_locations: Optional[Dict]):   # base_data_loader.py:64-139 (no ignore label)
        dim = len(shape)
        pad = [max(self.patch[d] - shape[d], 0) for d in range(dim)]
        lbs = [-pad[d] // 2 for d in range(dim)]
        ubs = [shape[d] + pad[d] // 2 + pad[d] % 2 - self.patch[d] for d in range(dim)]
        voxel = None
        if force_fg and class_locations:
            eligible = [k for k, v in class_locations.items() if len(v) > 0]
            if eligible:
                locs = class_locations[eligible[self.rs.choice(len(eligible))]]
                voxel = locs[self.rs.choice(len(locs))]
        if voxel is not None:
            lb = [max(lbs[d], int(voxel[d + 1]) - self.patch[d] // 2) for d in range(dim)]
        else:
            lb = [int(self.rs.randint(lbs[d], ubs[d] + 1)) for d in range(dim)]
        return lb, [lb[d] + self.patch[d] for d in range(dim)]

    def __iter__(self):
        return self

    def __next__(self):
        data_all = seg_all = None
        props, sel = [], [self.keys[i] for i in self.rs.choice(len(self.keys), self.B, replace=True)]
        for j, key in enumerate(sel):
            data, seg, pr = self.ds.load_case(key)
            props.append(pr)
            if data_all is None:
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
            for ax in self.mirror_axes:                        # MirrorTransform: each axis flipped with probability 1/2
                if self.rs.uniform() < 0.5:
                    data_all[j] = np.flip(data_all[j], ax + 1)
                    seg_all[j] = np.flip(seg_all[j], ax + 1)
        d = torch.from_numpy(np.ascontiguousarray(data_all))
        if self.pin and torch.cuda.is_available():
            d = d.pin_memory()
        return {"data": d, "seg": torch.from_numpy(np.ascontiguousarray(seg_all)), "properties": props, "keys": sel}


class DeviceFeed:
    """Host -> device hand-off of the batches (the reference's `inp.to(device, non_blocking=True)` on pinned batches,
    P/pretrain_AntoMask.py:343-345,390-392), double-buffered: batch k+1 is copied on a dedicated HIP stream into the other
    device slot while step k computes; `next(feed)` returns the device tensor of batch k after making the compute stream wait
    for its copy event.  Un-pinned host batches are first copied into a persistent pinned staging buffer of the slot."""

    def __init__(self, batches, device, key: str = "data", depth: int = 2):
        self.it, self.dev, self.key, self.depth = iter(batches), device, key, depth
        self.copy_stream = torch.cuda.Stream(device=device)
        self.dbuf, self.pin, self.ready = [None] * depth, [None] * depth, [None] * depth
        self.k = 0
        self._start(0)

    def _start(self, slot: int, after=None) -> bool:
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
        return True

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

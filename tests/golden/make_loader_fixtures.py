"""Generate tests/golden/loader_tiny.npz by RUNNING THE REFERENCE's data loader (build container only: needs /root/reference).

nnUNetDataLoader3D / nnUNetDataLoaderBase.get_bbox / nnUNetDataset (nnunetv2/training/dataloading/*.py) and get_patch_size
(nnunetv2/training/data_augmentation/compute_initial_patch_size.py) are imported from where they lie.  What is absent from the
image and therefore provided in memory:
  * batchgenerators (third party, `batchgenerators>=0.25` in the reference's pyproject.toml, not vendored): the `DataLoader` base
    class is restated from its published behaviour -- constructor arguments stored, `get_indices()` for infinite=True =
    np.random.choice(indices, batch_size, replace=True, p=sampling_probabilities) --, its file helpers are os.path / pickle
    one-liners, and augmentations.utils.rotate_coords_3d is coords @ Rx @ Ry @ Rz;
  * nnunetv2's LabelManager (pulls the whole planning stack): a two-attribute stand-in (all_labels, has_ignore_label).
The synthetic preprocessed folder comes from tests/helpers.make_synthetic_folder (data only: the fixture stores the selected
keys, the bounding boxes and checksums of the batches, not the folder)."""
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")


def _stub_modules():
    bg = types.ModuleType("batchgenerators")
    dl = types.ModuleType("batchgenerators.dataloading"); dlm = types.ModuleType("batchgenerators.dataloading.data_loader")

    class DataLoader:                      # batchgenerators.dataloading.data_loader.DataLoader, restated
        def __init__(self, data, batch_size, num_threads_in_multithreaded=1, seed_for_shuffle=None, return_incomplete=False, shuffle=True,
                     infinite=False, sampling_probabilities=None):
            self._data, self.batch_size, self.infinite, self.sampling_probabilities = data, batch_size, infinite, sampling_probabilities
            self.indices = None

        def get_indices(self):
            assert self.infinite
            return np.random.choice(self.indices, self.batch_size, replace=True, p=self.sampling_probabilities)

        def __next__(self):
            return self.generate_train_batch()
    dlm.DataLoader = DataLoader
    ut = types.ModuleType("batchgenerators.utilities"); ff = types.ModuleType("batchgenerators.utilities.file_and_folder_operations")
    from typing import List, Tuple, Union
    ff.join, ff.isfile, ff.isdir, ff.List, ff.Tuple, ff.Union = os.path.join, os.path.isfile, os.path.isdir, List, Tuple, Union
    ff.load_pickle = lambda f: pickle.load(open(f, "rb"))
    ff.subfiles = lambda folder, join=True, prefix=None, suffix=None, sort=True: sorted(
        (os.path.join(folder, i) if join else i) for i in os.listdir(folder)
        if os.path.isfile(os.path.join(folder, i)) and (suffix is None or i.endswith(suffix)) and (prefix is None or i.startswith(prefix)))
    ff.__all__ = ["join", "isfile", "isdir", "load_pickle", "subfiles", "List", "Tuple", "Union"]
    au = types.ModuleType("batchgenerators.augmentations"); auu = types.ModuleType("batchgenerators.augmentations.utils")

    def rotate_coords_3d(coords, ax, ay, az):
        rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
        ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
        rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
        return np.dot(coords.reshape(len(coords), -1).transpose(), rx @ ry @ rz).transpose().reshape(coords.shape)
    auu.rotate_coords_3d = rotate_coords_3d
    auu.rotate_coords_2d = lambda c, a: c
    lm = types.ModuleType("nnunetv2.utilities.label_handling.label_handling")
    lm.LabelManager = type("LabelManager", (), {})
    sys.modules.update({"batchgenerators": bg, "batchgenerators.dataloading": dl, "batchgenerators.dataloading.data_loader": dlm,
                        "batchgenerators.utilities": ut, "batchgenerators.utilities.file_and_folder_operations": ff,
                        "batchgenerators.augmentations": au, "batchgenerators.augmentations.utils": auu,
                        "nnunetv2.utilities.label_handling.label_handling": lm})


def main():
    _stub_modules()
    from nnunetv2.training.data_augmentation.compute_initial_patch_size import get_patch_size
    from nnunetv2.training.dataloading.data_loader_3d import nnUNetDataLoader3D
    from nnunetv2.training.dataloading.nnunet_dataset import nnUNetDataset
    from tests.helpers import checks, make_synthetic_folder
    out = {}
    rot = (-30. / 360 * 2. * np.pi, 30. / 360 * 2. * np.pi)
    for tag, fin in (("128", (128, 128, 128)), ("112", (112, 112, 128)), ("48", (48, 48, 48)), ("tiny", (24, 32, 24))):
        out[f"patchsize_{tag}"] = np.array([*fin, *get_patch_size(np.array(fin), rot, rot, rot, (0.85, 1.25))])
    label_manager = type("LM", (), {"all_labels": [1, 2], "has_ignore_label": False})()
    with tempfile.TemporaryDirectory() as td:
        make_synthetic_folder(td, 7)
        ds = nnUNetDataset(td, None, num_images_properties_loading_threshold=0)
        final = (24, 32, 24)
        initial = tuple(int(v) for v in get_patch_size(np.array(final), rot, rot, rot, (0.85, 1.25)))
        for run, (B, seed) in enumerate(((6, 101), (3, 202))):
            dl = nnUNetDataLoader3D(ds, B, initial, final, label_manager, oversample_foreground_percent=0.33, sampling_probabilities=None,
                                    pad_sides=None)
            np.random.seed(seed)
            bb = []
            orig = dl.get_bbox

            def spy(*a, _o=orig, **k):
                r = _o(*a, **k); bb.append(np.array(r)); return r
            dl.get_bbox = spy
            for it in range(3):
                b = dl.generate_train_batch()
                out[f"r{run}_b{it}_keys"] = np.array(list(b["keys"]))
                out[f"r{run}_b{it}_data_checks"] = np.stack([checks(__import__("torch").from_numpy(b["data"][j])) for j in range(B)])
                out[f"r{run}_b{it}_seg_sum"] = b["seg"].astype(np.int64).sum(axis=(1, 2, 3, 4))
                out[f"r{run}_b{it}_data_sample"] = b["data"].reshape(B, -1)[:, ::997].copy()
            out[f"r{run}_bbox"] = np.stack(bb)
            out[f"r{run}_cfg"] = np.array([B, seed, *initial, *final])
    np.savez_compressed(os.path.join(HERE, "loader_tiny.npz"), **out)
    print("loader_tiny.npz", os.path.getsize(os.path.join(HERE, "loader_tiny.npz")) // 1024, "KiB", out["patchsize_128"], out["r0_cfg"])


if __name__ == "__main__":
    main()

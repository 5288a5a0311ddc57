#!/usr/bin/env python
"""One-off image preprocessing for `--fine_tune` training (role of the reference's preprocess.py:10-45).

Every train2014 and val2014 JPEG is decoded, resized to 224 x 224 RGB (utils/image_utils.load_image) and stored as one
row of a single uint8 array; `./pickles/itoi.pickle` maps a file name to its row.  The reference keeps that array in an
HDF5 data set called "images"; when h5py is importable this script writes exactly that file and `Batch_Generator` reads it with
the reference's sorted fancy indexing; h5py is not part of this image, where the identical `(N, 224, 224, 3) uint8` array is
written as a memory-mappable `.npy` instead and opened wherever the reference opens the HDF5 file (`Parameters.hdf5_file`; a
`.hdf5` name resolves to the `.npy` next to it).

    python preprocess.py --coco_dir /data/coco --output_h5 /data/coco/train_val.npy
"""
import argparse
import os
import pickle
import sys
from glob import glob

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vae_captioning_amd.utils.image_utils import load_image  # noqa: E402

SIDE = 224


def image_files(coco_dir):
    """train2014 first, then val2014 -- the row order the index pickle records."""
    found = []
    for split in ("train2014", "val2014"):
        found += sorted(glob(os.path.join(coco_dir, "images", split, "*.jpg")))
    return found


def build(coco_dir, array_path, index_path="./pickles/itoi.pickle", report_every=1000):
    files = image_files(coco_dir)
    if not files:
        raise ValueError("no *.jpg under %s/images/train2014 or val2014" % coco_dir)
    h5 = None
    if array_path.endswith((".h5", ".hdf5")):  # the reference's container (preprocess.py:25-45): data set "images", needs h5py
        import h5py
        h5 = h5py.File(array_path, "w")
        rows = h5.create_dataset("images", (len(files), SIDE, SIDE, 3), dtype="uint8")
    else:
        rows = np.lib.format.open_memmap(array_path, mode="w+", dtype=np.uint8, shape=(len(files), SIDE, SIDE, 3))
    index = {}
    for row, path in enumerate(files):
        rows[row] = load_image(path, shape=(SIDE, SIDE))
        index[os.path.basename(path)] = row
        if row % report_every == 0:
            print("image %d of %d (%.1f %%)" % (row, len(files), 100.0 * row / len(files)))
    if h5 is not None:
        h5.close()
    else:
        rows.flush()
    os.makedirs(os.path.dirname(index_path) or ".", exist_ok=True)
    with open(index_path, "wb") as fh:
        pickle.dump(index, fh)
    print("wrote %s (%d images) and %s" % (array_path, len(files), index_path))
    return len(files)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--output_h5", default="train_val.npy", help="output image array (.npy; plays the reference's h5 file)")
    ap.add_argument("--coco_dir", required=True, help="MSCOCO directory (contains images/train2014, images/val2014)")
    a = ap.parse_args()
    target = a.output_h5
    if not target.endswith(".npy"):
        try:
            import h5py  # noqa: F401  (present: write the reference's HDF5 file under the name given)
        except ImportError:
            target = os.path.splitext(target)[0] + ".npy"
            print("h5py is not installed: writing %s (the same array, memory-mappable; Batch_Generator opens it in place of the .h5)" % target)
    build(a.coco_dir, target)

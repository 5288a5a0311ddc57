#!/usr/bin/env python
"""Resize every train2014 + val2014 image to 224x224 RGB once and store them in one array, with the
file-name -> row index map in ./pickles/itoi.pickle: the reference's preprocess.py:10-45.

    python preprocess.py --coco_dir /data/coco --output_h5 train_val.npy

The reference writes the array into an HDF5 dataset "images" (N, 224, 224, 3) uint8; h5py is not available
here, so the SAME array is written as a memory-mappable .npy (numpy's open_memmap).  `--fine_tune` training
reads it through `Parameters.hdf5_file` exactly where the reference reads the HDF5 file
(vae_captioning_amd/utils/batch_gen.py, open_image_array)."""
import argparse
import glob
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vae_captioning_amd.utils.image_utils import load_image  # noqa: E402


def main(params):
    coco_dir, out = params["coco_dir"], params["output_h5"]
    if not out.endswith(".npy"):
        out = os.path.splitext(out)[0] + ".npy"
    tr_files = sorted(glob.glob(coco_dir + "/images/train2014/*.jpg"))
    val_files = sorted(glob.glob(coco_dir + "/images/val2014/*jpg"))
    imgs = tr_files + val_files
    if len(imgs) == 0:
        raise ValueError("no images under %s/images/{train2014,val2014}" % coco_dir)
    N = len(imgs)
    dset = np.lib.format.open_memmap(out, mode="w+", dtype=np.uint8, shape=(N, 224, 224, 3))
    imtoi = {}
    for i, image_path in enumerate(imgs):
        dset[i] = load_image(image_path, shape=(224, 224))
        imtoi[image_path.split("/")[-1]] = i
        if i % 1000 == 0:
            print("processing %d/%d (%.2f%% done)" % (i, N, i * 100.0 / N))
    dset.flush()
    os.makedirs("./pickles", exist_ok=True)
    with open("./pickles/itoi.pickle", "wb") as wf:
        pickle.dump(obj=imtoi, file=wf)
        print("Saved image name to indices pickle")
    print("wrote ", out)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--output_h5", default="train_val.npy", help="output image array (.npy; the reference's h5 file)")
    parser.add_argument("--coco_dir", help="MSCOCO directory")
    main(vars(parser.parse_args()))

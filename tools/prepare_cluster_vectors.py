#!/usr/bin/env python
"""Cluster ("object") vectors c(I) for the GMM / AG priors: what the reference's two notebooks compute.

  train / val  (prepare_cluster_vectors_train_val.ipynb): from the MSCOCO instances_*.json annotations,
      c(I) = indicator over the 91 category ids (0..90) of the DISTINCT categories annotated in image I,
      normalised to sum to one -> ./obj_vectors/c_v.pickle  {file_name: float64[91]}
  test         (prepare_test_vectors.ipynb): from detector outputs {file_name: {'classes': [[...]], 'scores': [[...]]}}
      keep classes with score > 0.5, same normalisation -> ./obj_vectors/c_v_test.pickle
      (images with no detection above the threshold get the all-zero vector)

    python tools/prepare_cluster_vectors.py --instances a/instances_train2014.json a/instances_val2014.json
    python tools/prepare_cluster_vectors.py --test_scores ./obj_vectors/test_scores.pickle
"""
import argparse
import json
import os
import pickle
from collections import OrderedDict

import numpy as np

NUM_CLASSES = 90
INCLUDE_PROB = 0.5


def cluster_vectors_from_instances(j, class_num=NUM_CLASSES):
    """j: parsed instances_*.json -> {file_name: vector}; images without annotations are absent (as in the notebook)."""
    names = {img["id"]: img["file_name"] for img in j["images"]}
    cats = OrderedDict()
    for ann in j["annotations"]:
        lst = cats.setdefault(names[ann["image_id"]], [])
        if ann["category_id"] not in lst:
            lst.append(ann["category_id"])
    out = {}
    for fn, labels in cats.items():
        zv = np.zeros(class_num + 1)
        zv[labels] = 1
        out[fn] = zv / zv.sum()
    return out


def cluster_vectors_from_scores(test_scores, class_num=NUM_CLASSES, include_prob=INCLUDE_PROB):
    out = {}
    for imn, det in test_scores.items():
        v = np.zeros(class_num + 1)
        keep = np.argwhere(np.array(det["scores"][0]) > include_prob)[:, 0]
        v[np.array(det["classes"][0], dtype=int)[keep]] = 1
        out[imn] = v / v.sum() if v.sum() > 0 else v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", nargs="*", default=[], help="instances_train2014.json instances_val2014.json")
    ap.add_argument("--test_scores", default=None, help="pickle of detector outputs for the test images")
    ap.add_argument("--out_dir", default="./obj_vectors")
    a = ap.parse_args()
    os.makedirs(a.out_dir, exist_ok=True)
    if a.instances:
        c_v = {}
        for path in a.instances:
            with open(path) as rf:
                c_v.update(cluster_vectors_from_instances(json.load(rf)))
        with open(os.path.join(a.out_dir, "c_v.pickle"), "wb") as wf:
            pickle.dump(c_v, wf)
        print("wrote %d vectors to %s" % (len(c_v), os.path.join(a.out_dir, "c_v.pickle")))
    if a.test_scores:
        with open(a.test_scores, "rb") as rf:
            c_v = cluster_vectors_from_scores(pickle.load(rf))
        with open(os.path.join(a.out_dir, "c_v_test.pickle"), "wb") as wf:
            pickle.dump(c_v, wf)
        print("wrote %d vectors to %s" % (len(c_v), os.path.join(a.out_dir, "c_v_test.pickle")))


if __name__ == "__main__":
    main()

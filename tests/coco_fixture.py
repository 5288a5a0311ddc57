"""A miniature MSCOCO directory tree (images/{train,val,test}2014/*.jpg + annotations/*.json) built on the fly."""
import json
import os

import numpy as np

WORDS = ["a", "man", "dog", "cat", "on", "the", "beach", "red", "car", "sits", "runs", "with", "ball", "two"]


def build(root, n_train=6, n_val=4, n_test=2, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    root = str(root)
    if not root.endswith("/"):
        root += "/"
    os.makedirs(root + "annotations")
    iid = 100

    def split(name, n, with_caps):
        nonlocal iid
        os.makedirs(root + "images/%s2014" % name)
        images, anns = [], []
        for i in range(n):
            fn = "COCO_%s2014_%012d.jpg" % (name, iid)
            a = rng.integers(0, 256, size=(int(rng.integers(30, 60)), int(rng.integers(30, 60)), 3), dtype=np.uint8)
            Image.fromarray(a).save(root + "images/%s2014/%s" % (name, fn), quality=95)
            images.append(dict(id=iid, file_name=fn))
            if with_caps:
                for k in range(5):
                    ws = [WORDS[j] for j in rng.integers(0, len(WORDS), size=int(rng.integers(4, 9)))]
                    anns.append(dict(image_id=iid, id=iid * 10 + k, caption=" ".join(ws).capitalize() + "."))
            iid += 1
        return images, anns

    for name, n, caps, fn in (("train", n_train, True, "captions_train2014.json"), ("val", n_val, True, "captions_val2014.json"),
                              ("test", n_test, False, "image_info_test2014.json")):
        images, anns = split(name, n, caps)
        j = dict(images=images)
        if caps:
            j["annotations"] = anns
        with open(root + "annotations/" + fn, "w") as f:
            json.dump(j, f)
    return root


def vgg_weight_file(path, seed=4):
    """A random stand-in for vgg16_weights.npz: conv{i}_{j}_W/_b, fc6/fc7/fc8 _W/_b (sorted keys = load order)."""
    from vae_captioning_amd import spec
    PV = spec.init_vgg_params(seed=seed)
    out = {}
    for name, _ in spec.vgg_variables():
        layer, kind = name.split("/")[1], name.split("/")[2]
        key = {"fc1": "fc6", "fc2": "fc7"}.get(layer, layer) + ("_W" if kind.startswith("weights") else "_b")
        out[key] = PV[name]
    out["fc8_W"], out["fc8_b"] = np.zeros((4096, 1000), np.float32), np.zeros(1000, np.float32)
    np.savez(path, **out)
    return PV

"""MSCOCO front end: `Data` with the constructor arguments and methods of the reference's utils/data.py:16-175.

It resolves the data set's directory layout, builds the caption objects and the vocabulary, extracts VGG16 fc2 features
for a directory of images (cached as `./pickles/<split>.pickle`, the reference's `{file_name: float32 [1, 4096]}` format,
so files written by either side are interchangeable) and hands out the train / validation / test `Batch_Generator`s.

Feature extraction is the accelerated part: images are pushed through this build's VGG16 (trainer.VggEngine:
vc_conv3x3_fwd_f32, vc_maxpool2x2_fwd_f32, vc_gemm_f32) in batches, where the reference runs one `sess.run` per image
(utils/data.py:110-125)."""
import os
import pickle
from glob import glob

import numpy as np

from .batch_gen import Batch_Generator
from .captions import Captions, Dictionary
from .image_utils import load_image

_SPLITS = {"train": ("annotations/captions_train2014.json", "images/train2014/"),
           "valid": ("annotations/captions_val2014.json", "images/val2014/"),
           "test": ("annotations/image_info_test2014.json", "images/test2014/")}


class Data(object):
    def __init__(self, params, extract_features=False, weights_path=None, repartiton=False, gen_val_cap=None):
        self.params = params
        for split, (ann, img) in _SPLITS.items():  # train_cap_json / train_dir, valid_..., test_... (utils/data.py:21-28)
            setattr(self, split + "_cap_json", params.coco_dir + ann)
            setattr(self, split + "_dir", params.coco_dir + img)
        if repartiton and not gen_val_cap:
            raise ValueError("If using repartition must specify how many val images to use")
        self.repartiton, self.gen_val_cap = repartiton, gen_val_cap
        self.weights_path = weights_path
        self._vgg = None
        self.captions_tr = Captions(self.train_cap_json, params.cap_max_length)
        self.captions_val = Captions(self.valid_cap_json, params.cap_max_length)
        self.dictionary = Dictionary(self.captions_tr.captions, params.keep_words)  # vocabulary from the TRAIN captions only
        for caps in (self.captions_tr, self.captions_val):
            caps.index_captions(self.dictionary.word2idx)
        self.num_examples = self.captions_tr.num_captions
        self.train_feature_dict = None
        if extract_features:
            if not weights_path:
                raise ValueError("Specify imagenet weights path")
            self.train_feature_dict = self.extract_features_from_dir(self.train_dir)

    # ------------------------------------------------------------------ VGG16 fc2 features of a directory
    def _feature_extractor(self):
        if self._vgg is None:
            from ..trainer import VggEngine
            from .parameters import Parameters
            cfg = Parameters()
            cfg.mode, cfg.fine_tune = "inference", False  # vgg16(input_img) of the reference: no dropout, nothing trainable
            self._vgg = VggEngine(cfg)
            self._vgg.load_weights(self.weights_path)
        return self._vgg

    def extract_features_from_dir(self, data_dir, save_pickle=True, im_shape=(224, 224), batch=32):
        cache = os.path.join("./pickles", os.path.basename(os.path.dirname(data_dir)) + ".pickle")
        if os.path.exists(cache):
            print("Loading prepared feature vector from {}".format(cache))
            with open(cache, "rb") as fh:
                return pickle.load(fh)
        if not self.weights_path:
            raise ValueError("Specify imagenet weights path")
        print("Extracting features")
        import torch
        vgg = self._feature_extractor()
        files = sorted(glob(data_dir + "*.jpg"))
        features = {}
        for lo in range(0, len(files), batch):
            group = files[lo:lo + batch]
            pixels = np.stack([load_image(f, im_shape) for f in group]).astype(np.float32)
            fc2 = vgg.forward(torch.from_numpy(pixels).cuda()).cpu().numpy()
            features.update({os.path.basename(f): row[None].copy() for f, row in zip(group, fc2)})
        if save_pickle:
            os.makedirs("./pickles", exist_ok=True)
            with open(cache, "wb") as fh:
                pickle.dump(features, fh)
        return features

    # ------------------------------------------------------------------ batch generators
    def _image_source(self):
        return dict(use_hdf5=self.params.use_hdf5, hdf5_file=self.params.hdf5_file)

    def load_train_data_generator(self, batch_size, fine_tune=False, usehdf5=True, shard=None):
        """shard = (rank, world) (additive, data-parallel training): see batch_gen.shard_ranges."""
        from_images = fine_tune or not self.train_feature_dict
        kw = self._image_source() if from_images else {}
        kw["shard"] = shard
        self.train_batch_gen = Batch_Generator(self.train_dir, self.train_cap_json, self.captions_tr, batch_size,
                                               feature_dict=None if from_images else self.train_feature_dict, **kw)
        if self.repartiton:  # train on train2014 + val2014 minus the last gen_val_cap images (utils/batch_gen.py:71-96)
            self.train_batch_gen.repartiton(self.captions_val, self.extract_features_from_dir(self.valid_dir), self.gen_val_cap)
        return self.train_batch_gen

    def get_valid_data(self, val_batch_size=None, val_tr_unused=None, pretrained=True):
        feats = self.extract_features_from_dir(self.valid_dir) if pretrained else None
        self.valid_batch_gen = Batch_Generator(self.valid_dir, self.valid_cap_json, self.captions_val, val_batch_size, feature_dict=feats,
                                               get_image_ids=True, val_tr_unused=val_tr_unused, **self._image_source())
        return self.valid_batch_gen

    def get_test_data(self, test_batch_size=None, pretrained=True):
        feats = self.extract_features_from_dir(self.test_dir) if pretrained else None
        self.train_batch_gen = Batch_Generator(self.test_dir, train_cap_json=self.test_cap_json, batch_size=test_batch_size,
                                               feature_dict=feats, get_image_ids=True, get_test_ids=True)
        return self.train_batch_gen

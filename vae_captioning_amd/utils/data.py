"""`Data` with the constructor and methods of the reference's utils/data.py:16-175: MSCOCO directory layout,
caption objects + vocabulary, VGG16 fc2 feature extraction into ./pickles/<split>.pickle, and the
train / validation / test batch generators.

Feature extraction runs this build's VGG16 (trainer.VggEngine, vc_conv3x3_* / vc_gemm_f32) on batches of
images instead of one `sess.run` per image (utils/data.py:110-125); the pickle format is the reference's
({file_name: float32 [1, 4096]}), so files written by either side are interchangeable."""
import glob
import os
import pickle

import numpy as np

from .batch_gen import Batch_Generator
from .captions import Captions, Dictionary
from .image_utils import load_image


class Data(object):
    def __init__(self, params, extract_features=False, weights_path=None, repartiton=False, gen_val_cap=None):
        coco = params.coco_dir
        self.params = params
        self.train_cap_json = coco + "annotations/captions_train2014.json"
        self.valid_cap_json = coco + "annotations/captions_val2014.json"
        self.test_cap_json = coco + "annotations/image_info_test2014.json"
        self.train_dir = coco + "images/train2014/"
        self.valid_dir = coco + "images/val2014/"
        self.test_dir = coco + "images/test2014/"
        self.captions_tr = Captions(self.train_cap_json, params.cap_max_length)
        self.captions_val = Captions(self.valid_cap_json, params.cap_max_length)
        self.dictionary = Dictionary(self.captions_tr.captions, params.keep_words)
        self.captions_tr.index_captions(self.dictionary.word2idx)
        self.captions_val.index_captions(self.dictionary.word2idx)
        self.train_feature_dict = None
        self.num_examples = self.captions_tr.num_captions
        self.repartiton = repartiton
        self.gen_val_cap = gen_val_cap
        self.weights_path = weights_path
        self._vgg = None
        if repartiton and not gen_val_cap:
            raise ValueError("If using repartition must specify how many val images to use")
        if extract_features:
            if not weights_path:
                raise ValueError("Specify imagenet weights path")
            self.train_feature_dict = self.extract_features_from_dir(self.train_dir)

    # ------------------------------------------------------------------ features
    def _engine(self):
        if self._vgg is None:
            from ..trainer import VggEngine
            from .parameters import Parameters
            pv = Parameters()
            pv.mode, pv.fine_tune = "inference", False  # vgg16(input_img): no dropout, nothing trainable
            self._vgg = VggEngine(pv)
            self._vgg.load_weights(self.weights_path)
        return self._vgg

    def extract_features_from_dir(self, data_dir, save_pickle=True, im_shape=(224, 224), batch=32):
        """{file_name: fc2 [1, 4096]} for every *.jpg of data_dir; cached in ./pickles/<dir name>.pickle."""
        cache = "./pickles/" + data_dir.split("/")[-2] + ".pickle"
        if os.path.exists(cache):
            print("Loading prepared feature vector from {}".format(cache))
            with open(cache, "rb") as rf:
                return pickle.load(rf)
        print("Extracting features")
        if not self.weights_path:
            raise ValueError("Specify imagenet weights path")
        import torch
        vgg = self._engine()
        paths = sorted(glob.glob(data_dir + "*.jpg"))
        feature_dict = {}
        for s in range(0, len(paths), batch):
            chunk = paths[s:s + batch]
            imgs = np.stack([load_image(p, im_shape) for p in chunk]).astype(np.float32)
            fc2 = vgg.forward(torch.from_numpy(imgs).cuda()).cpu().numpy()
            for p, f in zip(chunk, fc2):
                feature_dict[p.split("/")[-1]] = f[None].copy()
        if save_pickle:
            os.makedirs("./pickles", exist_ok=True)
            with open(cache, "wb") as wf:
                pickle.dump(feature_dict, wf)
        return feature_dict

    # ------------------------------------------------------------------ generators
    def load_train_data_generator(self, batch_size, fine_tune=False, usehdf5=True):
        feature_dict = self.train_feature_dict
        val_cap = valid_feature_dict = None
        if self.repartiton:
            val_cap = self.captions_val
            valid_feature_dict = self.extract_features_from_dir(self.valid_dir)
        if fine_tune or not feature_dict:
            self.train_batch_gen = Batch_Generator(self.train_dir, self.train_cap_json, self.captions_tr, batch_size,
                                                   use_hdf5=self.params.use_hdf5, hdf5_file=self.params.hdf5_file, feature_dict=None)
        else:
            self.train_batch_gen = Batch_Generator(self.train_dir, self.train_cap_json, self.captions_tr, batch_size,
                                                   feature_dict=feature_dict)
        if self.repartiton:
            self.train_batch_gen.repartiton(val_cap, valid_feature_dict, self.gen_val_cap)
        return self.train_batch_gen

    def get_valid_data(self, val_batch_size=None, val_tr_unused=None, pretrained=True):
        valid_feature_dict = self.extract_features_from_dir(self.valid_dir) if pretrained else None
        self.valid_batch_gen = Batch_Generator(self.valid_dir, self.valid_cap_json, self.captions_val, val_batch_size,
                                               feature_dict=valid_feature_dict, get_image_ids=True, val_tr_unused=val_tr_unused,
                                               use_hdf5=self.params.use_hdf5, hdf5_file=self.params.hdf5_file)
        return self.valid_batch_gen

    def get_test_data(self, test_batch_size=None, pretrained=True):
        test_feature_dict = self.extract_features_from_dir(self.test_dir) if pretrained else None
        self.train_batch_gen = Batch_Generator(self.test_dir, train_cap_json=self.test_cap_json, batch_size=test_batch_size,
                                               feature_dict=test_feature_dict, get_image_ids=True, get_test_ids=True)
        return self.train_batch_gen

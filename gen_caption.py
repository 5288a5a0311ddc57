#!/usr/bin/env python
"""Caption one image: the command line and the `Generator` class of the reference's gen_caption.py
(:19-150) on the MI355X-native path.

    python gen_caption.py --img_path cat.jpg --checkpoint ./checkpoints/last_run.ckpt \\
        --params_path ./pickles/params_Normal_False_last_run_False.pickle --vocab_path ./pickles/capt_vocab.pickle \\
        [--gen_method greedy|beam_search|sample] [--beam_size 2] [--vgg_weights ./utils/vgg16_weights.npz]

Flow (gen_caption.py:73-130): load the pickled Parameters and the vocabulary, decode + resize the image,
VGG16 fc2 features [1, 4096], imf_emb -> decoder (prior z) -> greedy / beam search, print the caption.
Differences, all forced by what exists in this image:
  * the reference takes its features from Keras' ImageNet VGG16 (downloaded weights).  Here the features come
    from this build's VGG16 (`vc_conv3x3_*`) with the weights of `--vgg_weights` (the `vgg16_weights.npz` the
    training path uses, utils/image_embeddings.py:240-246) or, without that flag, the `cnn/*` variables of the
    checkpoint (present when the model was fine-tuned or `cnn` variables were saved, main.py:186-189).
  * `Dictionary(data_dict)` is called with keep_words from the params pickle (the reference call omits the
    argument and cannot run as written).
  * `--checkpoint` is a TF V2 checkpoint prefix (or an .npz archive written with --ckpt_format npz)."""
import argparse
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


class _ParamsUnpickler(pickle.Unpickler):
    """The reference pickles the Parameters INSTANCE (main.py:306-313), i.e. a reference to the class
    `utils.parameters.Parameters`; map it onto this build's class."""

    def find_class(self, module, name):
        if name == "Parameters" and module.endswith("parameters"):
            from vae_captioning_amd.utils.parameters import Parameters
            return Parameters
        return super().find_class(module, name)


class Generator(object):
    """Generate caption, given the image (gen_caption.py:19)."""

    def __init__(self, checkpoint_path, params_path, vocab_path, gen_method="greedy", vgg_weights=None):
        from vae_captioning_amd.utils.captions import Dictionary
        self.checkpoint_path = checkpoint_path
        self.params = self._load_params(params_path)
        self.gen_method = gen_method
        self.vgg_weights = vgg_weights
        if not vocab_path or not os.path.exists(vocab_path):
            raise ValueError("No caption vocabulary path specified, usually it can be found in the ./pickles folder "
                             "after model training")
        with open(vocab_path, "rb") as rf:
            data_dict = pickle.load(rf)
        self.data_dict = Dictionary(data_dict, getattr(self.params, "keep_words", 3))
        self.params.vocab_size = self.data_dict.vocab_size
        self._trainer = None
        self._vgg = None

    def _c_v_generator(self, image):
        # the reference leaves this unimplemented ("TODO: finish cluster vector implementation") and returns None
        return None

    def _load_params(self, params_path):
        """Load serialized Parameters class (gen_caption.py:50-55); a plain dict of attributes (what this
        build's main.py --save_params writes) is accepted too."""
        from vae_captioning_amd.utils.parameters import Parameters
        with open(params_path, "rb") as rf:
            obj = _ParamsUnpickler(rf).load()
        if isinstance(obj, dict):
            params = Parameters()
            for k, v in obj.items():
                setattr(params, k, v)
            return params
        return obj

    # ------------------------------------------------------------------ model pieces
    def _checkpoint_tensors(self):
        if self.checkpoint_path.endswith(".npz"):
            with np.load(self.checkpoint_path) as z:
                return {k: z[k] for k in z.files}
        from vae_captioning_amd import tf_bundle
        return tf_bundle.read_bundle(self.checkpoint_path)

    def _build(self):
        """imf_emb + Decoder (+ cv_emb) on restored variables (gen_caption.py:84-115)."""
        if self._trainer is not None:
            return
        from vae_captioning_amd import spec
        from vae_captioning_amd.trainer import Trainer, VggEngine
        from vae_captioning_amd.utils.parameters import Parameters
        p = self.params
        p.sample_gen = self.gen_method                     # gen_caption.py:83
        p.mode, p.fine_tune = "inference", False            # features are fed, as images_ps [None, 4096]
        tensors = self._checkpoint_tensors()
        tr = Trainer(p, p.vocab_size)
        tr.load_state_dict(tensors)
        p._vc_trainer = tr
        pv = Parameters()
        pv.mode, pv.fine_tune = "inference", False          # dropout_keep 1.0
        vgg = VggEngine(pv, lib=tr.lib)
        if self.vgg_weights:
            vgg.load_weights(self.vgg_weights)
        elif all(n in tensors for n, _ in spec.vgg_variables()):
            vgg.load_params(tensors)
        else:
            raise ValueError("no VGG16 weights: give --vgg_weights vgg16_weights.npz (the checkpoint holds no cnn/* variables)")
        self._trainer, self._vgg = tr, vgg

    def _get_features(self, img_path):
        """Loads image, extracts fc2 features -> ([1, 4096] float32, PIL image)  (gen_caption.py:57-71)."""
        import torch
        from vae_captioning_amd.utils.image_utils import keras_load_img
        self._build()
        x, img = keras_load_img(img_path, target_size=(224, 224))
        fc2 = self._vgg.forward(torch.from_numpy(x).cuda())
        return fc2.cpu().numpy(), img

    def generate_caption(self, img_path, beam_size=2):
        """-> [{'image_id': file name, 'caption': text}]  (gen_caption.py:73-130)."""
        from vae_captioning_amd.vae_model.decoder import Decoder
        if not img_path or not os.path.exists(img_path):
            raise ValueError("Image not found")
        self._build()
        decoder = Decoder(None, None, None, self.params, self.data_dict)
        im_id = [img_path.split("/")[-1]]
        feature_vector, image = self._get_features(img_path)
        c_v = self._c_v_generator(image) if self.params.use_c_v else None
        if self.gen_method == "beam_search":
            return decoder.beam_search(None, im_id, feature_vector, None, c_v, beam_size=int(beam_size))
        if self.gen_method in ("greedy", "sample"):
            sent, _ = decoder.online_inference(None, im_id, feature_vector, None, c_v=c_v)
            return sent
        raise ValueError("gen_method must be greedy, beam_search or sample")


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Specify generation parameters")
    parser.add_argument("--img_path", help="Path to the image")
    parser.add_argument("--checkpoint", help="Model checkpoint path")
    parser.add_argument("--vocab_path", default="./pickles/capt_vocab.pickle", help="Indices to words dictionary")
    parser.add_argument("--gpu", default="", help="Specify GPU number if use GPU")
    parser.add_argument("--c_v_generator", default=None, help="If use cluster vectors, specify tensorflow api model (unused, as in the reference)")
    parser.add_argument("--gen_method", default="greedy", help="greedy, beam_search or sample")
    parser.add_argument("--params_path", default=None, help="specify params pickle file")
    parser.add_argument("--beam_size", default=2, help="If using beam_search, specify beam_size")
    parser.add_argument("--vgg_weights", default=None, help="vgg16_weights.npz for the feature extractor (additive flag)")
    args = parser.parse_args()
    if args.gpu != "":
        os.environ["HIP_VISIBLE_DEVICES"] = args.gpu
    generator = Generator(checkpoint_path=args.checkpoint, params_path=args.params_path, vocab_path=args.vocab_path,
                          gen_method=args.gen_method, vgg_weights=args.vgg_weights)
    caption = generator.generate_caption(args.img_path, args.beam_size)
    print(caption[0]["caption"])

"""`vgg16` with the constructor and attributes of utils/image_embeddings.py:14-246, executed
eagerly on the GPU by trainer.VggEngine (conv3x3+bias+ReLU x13, max-pool x5, fc1/fc2 via libvaecap)."""
import numpy as np
import torch

from .. import spec
from ..trainer import VggEngine
from .parameters import Parameters


class vgg16(object):
    def __init__(self, imgs, weight_file=None, sess=None, trainable_fe=False, trainable_top=False, dropout_keep=1.0,
                 engine=None, params=None):
        """imgs: [B, 224, 224, 3] float RGB 0..255 (numpy or device tensor).  `sess` is accepted for
        signature compatibility and ignored (there is no session: execution is eager).  params: build on the VGG16 of that
        Parameters object's session (main.py:74-78 inside the training graph): `.fc2` then runs the session's engine with its step
        counter (dropout stream) and the images become part of the step's batch."""
        self.imgs = imgs
        self._session = None
        if params is not None:
            from .. import session
            tr = session.get(params)
            if tr.vgg is None:
                raise ValueError("vgg16(params=...): the session has no VGG16 (params.fine_tune is off)")
            engine = tr.vgg
            self._session = tr
            session.stage(params, owner='vgg16', images=imgs)
        self.dropout_keep = dropout_keep
        self.trainable_fe = trainable_fe
        self.trainable_top = trainable_top
        if engine is None:
            p = Parameters()
            p.fine_tune = bool(trainable_fe or trainable_top)
            p.cnn_dropout = dropout_keep
            engine = VggEngine(p)
        self.engine = engine
        # the 30 variables in construction order (= the order load_weights assigns them, Q18)
        self.parameters = [engine.store.param(n) for n in engine.store.names()]
        if weight_file is not None:
            self.load_weights(weight_file, sess)

    def _images(self, imgs):
        if isinstance(imgs, torch.Tensor):
            return imgs.to("cuda", torch.float32)
        return torch.from_numpy(np.ascontiguousarray(imgs, dtype=np.float32)).cuda()

    def forward(self, imgs=None, step=None):
        if step is None and self._session is not None:
            step = self._session.cap.step
        return self.engine.forward(self._images(self.imgs if imgs is None else imgs), step)

    @property
    def fc2(self):
        """fc2 features [B, 4096] of `self.imgs` (image_embeddings.py:229-238)."""
        return self.forward()

    def load_weights(self, weight_file, sess=None):
        self.engine.load_weights(weight_file)

    @staticmethod
    def variable_names():
        return [n for n, _ in spec.vgg_variables()]

"""Shared execution context behind the reference-named facades.

The reference builds ONE TensorFlow graph and runs it through a `tf.Session`; its classes
(vgg16, Encoder, Decoder, the optimiser functions) find each other through that graph and its
variable scopes.  Here the equivalent shared object is a `Trainer` (flat parameter store +
workspaces) attached to the `Parameters` instance every facade receives."""
from .trainer import Trainer


def get(params, vocab=None):
    """The Trainer that belongs to this Parameters object (created on first use)."""
    tr = getattr(params, "_vc_trainer", None)
    if tr is None:
        v = vocab or params.vocab_size
        if v is None:
            raise ValueError("params.vocab_size must be set before building the model (main.py:92)")
        tr = Trainer(params, int(v))
        params._vc_trainer = tr
    return tr

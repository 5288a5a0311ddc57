"""Shared execution context behind the reference-named facades.

The reference builds ONE TensorFlow graph and runs it through a `tf.Session`; its classes
(vgg16, Encoder, Decoder, the optimiser functions) find each other through that graph and its
variable scopes, and the arrays of a step arrive through `feed_dict` at `sess.run` time.  Here the
equivalent shared object is a `Trainer` (flat parameter store + workspaces) attached to the
`Parameters` instance every facade receives, and the arrays arrive through the facades'
CONSTRUCTOR ARGUMENTS (the positions the reference's placeholders occupy, main.py:43-60,99-102):
each facade stages what it was given, and the first graph stage that executes (`Encoder.q_net`, or
`Decoder.px_z_fi` under --no_encoder) uploads the staged batch -- `bind` below.  Facades built with
`None` arguments keep the old contract (the caller has called `Trainer.set_batch` itself)."""
import numpy as np

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


class Staged(object):
    """What `layers.dense(features, ..., name='imf_emb' | 'cv_emb')` returns: the facade-side stand-in for the symbolic tensor
    (main.py:94,108).  The product is computed inside the engine's fused input buffer once the whole batch is known."""

    def __init__(self, name, source):
        self.name, self.source = name, source


def stage(params, owner=None, **arrays):
    """Record the arrays a facade was constructed with.  `owner` names the facade ("encoder", "decoder", "imf_emb", ...): a None
    argument forgets only what THE SAME owner staged under that name earlier -- rebuilding one facade with None never erases what
    another facade staged under a shared key (Encoder and Decoder both stage `lengths`; `c_v` comes from three places)."""
    st = params.__dict__.setdefault("_vc_staged", {})
    who = params.__dict__.setdefault("_vc_staged_by", {})
    for k, v in arrays.items():
        if v is not None:
            st[k] = v
            who[k] = owner
        elif k in st and who.get(k) == owner:
            st.pop(k)
            who.pop(k, None)
    return st


def staged(params):
    return params.__dict__.get("_vc_staged", {})


def bind(params):
    """Make the staged arrays the engine's resident batch: builds the Trainer.set_batch dict from what Encoder / Decoder / layers.dense
    / vgg16 were given and uploads it -- ALWAYS: the copy is one asynchronous H2D transfer, and neither object identity (a caller may
    refill a preallocated array in place) nor a remembered earlier upload (Trainer.set_batch may have replaced it since) says what
    is resident.  Returns the device feature tensor the step must use (fc2 of the session's VGG16 when fine-tuning) or None
    (uploaded precomputed features).  Raises if the staged set is incomplete or inconsistent -- a step never silently trains on a
    stale batch."""
    tr = get(params)
    st = staged(params)
    if not st:
        return None
    cap = tr.cap
    need = ["cap_dec", "lengths"] + ([] if params.no_encoder else ["cap_enc"])
    missing = [k for k in need if k not in st]
    if missing:
        raise ValueError("facade inputs incomplete: %s not given (Encoder(images_fv, cap_enc, lengths, params) and "
                         "Decoder(images_fv, cap_dec, lengths, params, data_dict) must both receive arrays, main.py:99-102)" % missing)
    cap_dec = np.asarray(st["cap_dec"], np.int32)
    # under --no_encoder the labels are still the <EOS>-shifted captions (main.py:153): Decoder callers stage them as `cap_enc` too
    if "cap_enc" not in st:
        raise ValueError("facade inputs incomplete: the labels (cap_enc, main.py:153) were not given; stage them with "
                         "session.stage(params, cap_enc=...) when no Encoder is built")
    cap_enc = np.asarray(st["cap_enc"], np.int32)
    if cap_dec.shape != cap_enc.shape:
        raise ValueError("Encoder captions %s and Decoder captions %s differ in shape" % (cap_enc.shape, cap_dec.shape))
    batch = dict(cap_dec=cap_dec, cap_enc=cap_enc, lengths=np.asarray(st["lengths"], np.int32))
    nc = params.num_captions if params.mode == "training" else 1
    N = cap_dec.shape[0]
    feats_dev = None
    src = st.get("features")
    if src is None:
        raise ValueError("facade inputs incomplete: images_fv must come from layers.dense(features, embed_size, name='imf_emb', params=params)")
    import torch
    if isinstance(src, torch.Tensor):
        feats_dev = src                      # fc2 of the session's VGG16 (fine-tuning)
    else:
        f = np.asarray(src, np.float32)
        if f.shape[0] == N and nc > 1:       # the reference tiles the rows x nc before imf_emb (main.py:84-89): undo, checking it IS a tiling
            t = f.reshape(N // nc, nc, -1)
            if not np.array_equal(t, np.repeat(t[:, :1], nc, axis=1)):
                raise ValueError("features have one row per caption but are not a x%d tiling of per-image rows (main.py:84-89)" % nc)
            f = np.ascontiguousarray(t[:, 0])
        if f.shape[0] * nc != N:
            raise ValueError("features have %d rows for %d caption rows at num_captions=%d" % (f.shape[0], N, nc))
        batch["features"] = f
    if cap.use_ci:
        cv = st.get("c_v")
        if cv is None:
            raise ValueError("facade inputs incomplete: the cluster vectors were not given (decoder.c_i_ph / encoder.c_i_ph, main.py:108-116)")
        batch["c_v"] = np.asarray(cv, np.float32)
    if feats_dev is not None:   # the images are on the device already (vgg16 facade): only the caption side is uploaded
        tr.cap.set_batch(batch)
    else:
        tr.set_batch(batch)
    return feats_dev

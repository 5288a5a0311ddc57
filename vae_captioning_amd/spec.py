"""Variable table: the checkpoint names and shapes of every trainable variable of
the reference graph (main.py:186-191), kept in ONE
place so the TF scope strings can be corrected against a real checkpoint later.

The exact RNN scope strings depend on the TF minor version (TF-sem., unverified).
"""
import numpy as np

NUM_CLUSTERS = 90
ENC_CELL = "encoder/multi_rnn_cell/cell_0/lstm_cell/"
DEC_CELL = "decoder/net/multi_rnn_cell/cell_0/lstm_cell/"

VGG_CONV = [("conv1_1", 3, 64), ("conv1_2", 64, 64),
            ("conv2_1", 64, 128), ("conv2_2", 128, 128),
            ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256),
            ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512),
            ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512)]
# a maxpool follows these layers (utils/image_embeddings.py:59,88,128,168,208)
VGG_POOL_AFTER = {"conv1_2", "conv2_2", "conv3_3", "conv4_3", "conv5_3"}


def uses_ci(p):
    """main.py:52-53,103-104: cv_emb exists with --c_v or a GMM/AG prior."""
    return bool(p.use_c_v) or p.prior in ("GMM", "AG")


def head_scope(prior, k):
    return ("encoder/gmm_ll_%d/" if prior == "GMM" else "encoder/ag_ll_%d/") % k


def vgg_var_names(layer):
    """conv5_x variables are named weights_conv / biases_conv
    (utils/image_embeddings.py:176-201)."""
    if layer.startswith("conv5"):
        return "cnn/%s/weights_conv" % layer, "cnn/%s/biases_conv" % layer
    return "cnn/%s/weights" % layer, "cnn/%s/biases" % layer


def caption_variables(p, vocab):
    """Ordered [(name, shape)] of the non-CNN variables, in the order
    ops/optimizers.py:4-12 collects them: cv_emb, imf_emb, decoder/*, encoder/*."""
    E, He, Hd, L, S = p.embed_size, p.encoder_hidden, p.decoder_hidden, p.latent_size, p.gen_z_samples
    v = []
    if uses_ci(p):
        v += [("cv_emb/kernel", (NUM_CLUSTERS, E)), ("cv_emb/bias", (E,))]
    v += [("imf_emb/kernel", (p.cnn_feature_size, E)), ("imf_emb/bias", (E,))]
    v += [("decoder/net/dec_embeddings", (vocab, E)),
          (DEC_CELL + "kernel", (E + Hd, 4 * Hd)), (DEC_CELL + "bias", (4 * Hd,))]
    if not p.no_encoder:
        v += [("decoder/net/z_rnn/kernel", (S * L, E)), ("decoder/net/z_rnn/bias", (E,))]
    v += [("decoder/rnn_logits/kernel", (Hd, vocab)), ("decoder/rnn_logits/bias", (vocab,))]
    if not p.no_encoder:
        v += [("encoder/enc_embeddings", (vocab, E)),
              (ENC_CELL + "kernel", (E + He, 4 * He)), (ENC_CELL + "bias", (4 * He,))]
        if p.prior == "Normal":
            v += [("encoder/dense/kernel", (He, L)), ("encoder/dense/bias", (L,)),
                  ("encoder/dense_1/kernel", (He, L)), ("encoder/dense_1/bias", (L,))]
        else:
            for k in range(NUM_CLUSTERS):
                s = head_scope(p.prior, k)
                v += [(s + "dense/kernel", (He, L)), (s + "dense/bias", (L,)),
                      (s + "dense_1/kernel", (He, L)), (s + "dense_1/bias", (L,))]
    return v


def vgg_variables():
    """The 30 cnn/* variables in construction order = vgg16.parameters order
    (utils/image_embeddings.py:36-238), which is also the npz load order."""
    v = []
    for name, ci, co in VGG_CONV:
        wn, bn = vgg_var_names(name)
        v += [(wn, (3, 3, ci, co)), (bn, (co,))]
    v += [("cnn/fc1/weights", (25088, 4096)), ("cnn/fc1/biases", (4096,)),
          ("cnn/fc2/weights", (4096, 4096)), ("cnn/fc2/biases", (4096,))]
    return v


def _glorot(rng, shape):
    fan_in = int(np.prod(shape[:-1])) if len(shape) > 1 else shape[0]
    fan_out = shape[-1]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def init_caption_params(p, vocab, seed=0):
    """TF default initialisers (TF-sem.): glorot_uniform kernels/embeddings, zero
    biases.  Seeded numpy stream; TF's Philox stream cannot be matched."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in caption_variables(p, vocab):
        if name.endswith("bias"):
            out[name] = np.zeros(shape, np.float32)
        else:
            out[name] = _glorot(rng, shape)
    return out


def init_vgg_params(seed=0):
    """Random stand-in for the ImageNet weights (no network here): He-normal."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in vgg_variables():
        if len(shape) == 1:
            out[name] = np.zeros(shape, np.float32)
        else:
            fan_in = int(np.prod(shape[:-1]))
            out[name] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in)))
    return out

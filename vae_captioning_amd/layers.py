"""`layers.dense` at the two call sites of the reference's graph builder that are not inside its own modules:
`images_fv = layers.dense(features, params.embed_size, name='imf_emb')` (main.py:94) and
`c_i_emb = layers.dense(cl_vectors, params.embed_size, name='cv_emb')` (main.py:108).  The products themselves run inside the
engine (imf_emb on the B image rows, tiled x num_captions into the LSTM input buffer; cv_emb straight into its init-chain slot), so the
call stages its input array for the session and returns the stand-in the Encoder / Decoder constructors accept."""
from . import session


def dense(inputs, units, name=None, params=None):
    if params is None:
        raise TypeError("layers.dense needs params= (the Parameters object that carries the session)")
    if name == "imf_emb":
        if units != params.embed_size:
            raise ValueError("imf_emb has %d units in this model, not %d" % (params.embed_size, units))
        session.stage(params, owner='imf_emb', features=inputs)
        return session.Staged("imf_emb", inputs)
    if name == "cv_emb":
        if units != params.embed_size:
            raise ValueError("cv_emb has %d units in this model, not %d" % (params.embed_size, units))
        session.stage(params, owner='cv_emb', c_v=inputs)
        return session.Staged("cv_emb", inputs)
    raise ValueError("layers.dense(name=%r): only the graph builder's own two layers ('imf_emb', 'cv_emb') exist outside the modules" % name)

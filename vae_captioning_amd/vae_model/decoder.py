"""`Decoder` with the constructor, attributes and methods of vae_model/decoder.py:10-320:
px_z_fi (training graph), online_inference (greedy / sampling) and beam_search, executed on the GPU
by engine.CaptionEngine and generate.CaptionGenerator (all images and beams batched per step)."""
import numpy as np

from .. import session, spec
from ..generate import CaptionGenerator


class Decoder(object):
    def __init__(self, images_fv, captions, lengths, params, data_dict):
        """images_fv: layers.dense(..., name='imf_emb'); captions: cap_dec [N, T] int (`<BOS>...`, 0-padded); lengths [N]
        (vae_model/decoder.py:13-20).  Arrays given here are what the step runs on (session.bind)."""
        from .encoder import check_images_fv
        check_images_fv(images_fv)
        session.stage(params, owner='decoder', cap_dec=captions, lengths=lengths)
        self.images_fv = images_fv
        self.captions = captions
        self.lengths = lengths
        self.params = params
        self.data_dict = data_dict  # needs .word2idx / .idx2word / .vocab_size
        self.c_i = None
        self.c_i_ph = None
        self.cap_clusters = None

    def px_z_fi(self, observed, gen_mode=False):
        """Training graph: returns (model, x_logits, shpe, (initial_state, final_state, sample)) like
        decoder.py:143.  `observed` = {'z': qz} or {} (--no_encoder); the z actually used is the
        encoder's sample held by the shared engine (zs.BayesianNet(observed) semantics)."""
        if gen_mode:
            raise ValueError("generation runs through online_inference / beam_search")
        tr = session.get(self.params)
        eng = tr.cap
        if session.staged(self.params) and not eng.enc:   # --no_encoder: no q_net ran, this is the first stage of the step
            if self.c_i_ph is not None:
                session.stage(self.params, owner='decoder', c_v=self.c_i_ph)
            feats = session.bind(self.params)
            if tr.vgg is not None and tr.vgg.wd:
                tr.vgg.reg_sumsq(eng.red.data_ptr() + 12)
            eng.fw_prepare(feats)
        logits = eng.fw_decode()
        nid = eng.n_init_d
        hs, cs = eng.buf["hs_d"], eng.buf["cs_d"]
        shpe = (tuple(eng.buf["z"].shape) if eng.enc else (), (eng.T * eng.N, self.params.decoder_hidden),
                (eng.N, eng.T, self.params.decoder_hidden))
        return None, logits, shpe, ((cs[nid], hs[nid]), (cs[-1], hs[-1]), None)

    # ------------------------------------------------------------------ generation
    def _gen(self):
        tr = session.get(self.params)
        g = getattr(tr, "_generator", None)
        if g is None:
            g = tr._generator = CaptionGenerator(tr.cap)
        return g

    def _features(self, in_pictures):
        tr = session.get(self.params)
        a = np.asarray(in_pictures, np.float32)
        if a.ndim == 4:  # raw images: run the fine-tuned feature extractor (ops/inference.py:9-12)
            import torch
            return tr.vgg.forward(torch.from_numpy(a).cuda())
        return a.reshape(a.shape[0], -1)

    def online_inference(self, sess, picture_ids, in_pictures, image_f_inputs, stop_word="<EOS>", c_v=None):
        """decoder.py:145-201.  Returns (cap_list, cap_raw)."""
        d = self.data_dict
        bos, eos = d.word2idx["<BOS>"], d.word2idx[stop_word]
        use_cv = c_v if (spec.uses_ci(self.params) and c_v is not None and len(c_v)) else None
        if self.params.sample_gen == "sample":  # tf.multinomial(logits / temperature): same distribution, own Philox stream
            raw = self._gen().sample(self._features(in_pictures), use_cv, None, bos, eos, self.params.gen_max_len)
        else:
            raw = self._gen().greedy(self._features(in_pictures), use_cv, None, bos, eos, self.params.gen_max_len)
        cap_list = []
        for pid, toks in zip(picture_ids, raw):
            words = [d.idx2word[t] for t in toks if t not in (bos, eos)]
            cap_list.append({"image_id": pid, "caption": " ".join(words)})
        return cap_list, raw

    def beam_search(self, sess, picture_ids, in_pictures, image_f_inputs, c_v=None, beam_size=2, ret_beams=False,
                    len_norm_f=0.7):
        """decoder.py:203-320.  Returns cap_list."""
        d = self.data_dict
        bos, eos = d.word2idx["<BOS>"], d.word2idx["<EOS>"]
        use_cv = c_v if (spec.uses_ci(self.params) and c_v is not None and len(c_v)) else None
        res = self._gen().beam_search(self._features(in_pictures), use_cv, None, bos, eos, beam_size,
                                      self.params.gen_max_len, len_norm_f)
        cap_list = []
        for pid, beams in zip(picture_ids, res):
            texts = [" ".join(d.idx2word[t] for t in s if t not in (bos, eos)) for s, _ in beams]
            cap_list.append({"image_id": pid, "caption": texts if ret_beams else texts[0]})
        return cap_list

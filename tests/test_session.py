"""Host logic of the facade session (no GPU): which arrays a step trains on when facades are rebuilt."""
import numpy as np

from vae_captioning_amd import session


class _P(object):
    pass


def test_none_forgets_only_what_the_same_facade_staged():
    p = _P()
    lens, cap = np.arange(4), np.zeros((4, 3), np.int32)
    session.stage(p, owner="encoder", cap_enc=cap, lengths=lens)
    session.stage(p, owner="decoder", cap_dec=cap, lengths=None)      # Decoder(fv, cap_dec, None, ...): the Encoder's lengths stay
    assert session.staged(p)["lengths"] is lens and set(session.staged(p)) == {"cap_enc", "cap_dec", "lengths"}
    session.stage(p, owner="encoder", cap_enc=None, lengths=None)     # the owner itself rebuilt with None: forgotten
    assert set(session.staged(p)) == {"cap_dec"}
    session.stage(p, owner="decoder", lengths=lens)
    session.stage(p, owner="encoder", lengths=lens + 1)               # the last facade given an array wins ...
    assert session.staged(p)["lengths"][0] == 1
    session.stage(p, owner="decoder", lengths=None)                   # ... and only that facade can withdraw it
    assert "lengths" in session.staged(p)
    session.stage(p, owner="encoder", lengths=None)
    assert "lengths" not in session.staged(p)

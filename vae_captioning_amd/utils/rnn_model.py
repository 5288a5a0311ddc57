"""make_rnn_cell / rnn_placeholders with the signatures of utils/rnn_model.py:7-51.

The returned cell is an eager single-layer LSTM (the reference only ever builds one layer:
encoder_rnn_layers = decoder_rnn_layers = 1, utils/parameters.py:20,25) whose step is the fused
MFMA kernel `vc_lstm_step_fwd_f32`; output dropout (DropoutWrapper(output_keep_prob)) is applied
to the returned output only, never to the carried state."""
import collections

import numpy as np
import torch

from .. import abi
from ..abi import ptr as P

LSTMStateTuple = collections.namedtuple("LSTMStateTuple", ("c", "h"))


def rnn_placeholders(state):
    """The reference turns the state tensors into placeholders-with-default so that a decode step can
    be fed the previous state (utils/rnn_model.py:7-21).  Eager execution needs no placeholder: the
    state tuple itself is what the caller passes back in."""
    if isinstance(state, LSTMStateTuple) or isinstance(state, torch.Tensor):
        return state
    return tuple(rnn_placeholders(s) for s in state)


class LSTMStack(object):
    def __init__(self, num_units, dropout_keep_prob=1.0):
        self.num_units = num_units
        self.output_size = num_units
        self.state_size = (LSTMStateTuple(num_units, num_units),)
        self.keep = dropout_keep_prob
        self.kernel = None  # [E + H, 4H]; bound by the model or created on first call
        self.bias = None
        self.lib = abi.load()

    def bind(self, kernel, bias):
        self.kernel, self.bias = kernel, bias
        return self

    def zero_state(self, batch_size, dtype=None):
        z = lambda: torch.zeros((batch_size, self.num_units), dtype=torch.float32, device="cuda")
        return (LSTMStateTuple(z(), z()),)

    def __call__(self, inputs, state, drop_mask=None):
        (c, h), = state
        N, E = inputs.shape
        H = self.num_units
        if self.kernel is None:  # TF default initialiser: glorot_uniform kernel, zero bias
            lim = np.sqrt(6.0 / (E + H + 4 * H))
            self.kernel = (torch.rand((E + H, 4 * H), device="cuda") * 2 - 1) * lim
            self.bias = torch.zeros(4 * H, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        gact = torch.empty((N, 4 * H), dtype=torch.float32, device="cuda")
        ws = torch.empty(max(self.lib.vc_gemm_workspace_bytes(N, 4 * H, E), 16) // 4 + 4, device="cuda")
        self.lib.vc_gemm_f32(st, 0, 0, N, 4 * H, E, P(inputs), E, P(self.kernel), 4 * H, P(gact), 4 * H, P(self.bias), 0, P(ws), ws.numel() * 4)
        c2, h2 = torch.empty_like(c), torch.empty_like(h)
        ones = torch.ones((N,), dtype=torch.int32, device="cuda")
        self.lib.vc_lstm_step_fwd_f32(st, N, H, 0, P(h), P(c), self.kernel.data_ptr() + E * 4 * H * 4, P(gact), P(ones), P(c2), P(h2))
        out = h2
        if self.keep < 1 and drop_mask is not None:
            out = torch.empty_like(h2)
            self.lib.vc_dropout_f32(st, P(h2), P(drop_mask), self.keep, N * H, P(out))
        return out, (LSTMStateTuple(c2, h2),)


def make_rnn_cell(rnn_layer_sizes, dropout_keep_prob=1.0, attn_length=0, base_cell=None):
    if len(rnn_layer_sizes) != 1 or attn_length:
        raise NotImplementedError("the reference only builds single-layer cells without attention")
    return LSTMStack(int(rnn_layer_sizes[0]), dropout_keep_prob)

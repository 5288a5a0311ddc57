"""non_cnn_optimizer / cnn_optimizer with the signatures of ops/optimizers.py:3-82.

The reference returns graph ops; here `optimize` is a callable that runs backward + the update on
the shared engine, `global_step` / `global_norm` are device tensors that it refreshes."""
from .. import session


def non_cnn_optimizer(loss, params):
    """Variables cv_emb + imf_emb + decoder/* (+ encoder/*), tf.gradients, clip_by_global_norm(5.0),
    Adam(lr, beta1=0.8) / SGD / Momentum(0.9) with staircase decay (ops/optimizers.py:3-47)."""
    tr = session.get(params)
    cap = tr.cap

    def optimize():
        dfe = cap.backward(want_dfeatures=tr.vgg is not None)
        cap.join_off_chain()   # (Trainer.train_step keeps the weight gradients on their own stream up to the optimiser; here: program order)
        tr._pending_dfeatures = dfe
        cap.pack_tail()
        if tr.vgg is None or not tr.vgg.train:
            tr.all_reduce_grads()
            cap.apply_gradients()
        return 0.0

    return optimize, cap.step, cap.ns[0:1]


def cnn_optimizer(loss, params):
    """All trainable cnn/* variables, no clipping, Adam(cnn_lr, beta1=0.8) by default
    (ops/optimizers.py:49-82).  Must run after non_cnn_optimizer's `optimize` in the same step."""
    tr = session.get(params)

    def optimize():
        tr.vgg.backward(tr._pending_dfeatures)
        tr.all_reduce_grads()
        tr.cap.apply_gradients()
        tr.vgg.apply_gradients(tr.cap.scal)
        return 0.0

    return optimize, tr.cap.step

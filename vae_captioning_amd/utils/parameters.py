"""Run configuration object with the attribute names, default values and command-line flags of the reference's
``Parameters`` (utils/parameters.py:2-66 attributes, :75-132 flags), so that ``main.py`` invocations written for the
reference keep working.  The flag table below is the single place that maps a flag to its attribute.

Differences: ``--gpu`` selects devices through HIP_VISIBLE_DEVICES (a comma list is allowed
for data-parallel runs) and is optional; a handful of additive flags (``--synthetic``, ``--ckpt_format`` ...) exist only here.
"""
import argparse
import os

_COCO = "/home/luoyy16/datasets-large/mscoco/coco/"

# attribute -> default.  Grouped as in the reference: model sizes, decoding, regularisation / optimiser, bookkeeping,
# fine-tuning, data preparation.
_DEFAULTS = dict(
    latent_size=150, num_clusters=90, num_epochs=20, learning_rate=0.0005, num_captions=5, batch_size=32, cnn_feature_size=4096,
    temperature=1.0, sample_gen="beam_search", beam_size=10, gen_max_len=30, gen_z_samples=100,
    encoder_rnn_layers=1, encoder_hidden=512, decoder_rnn_layers=1, decoder_hidden=512, embed_size=256, std=0.1,
    dec_keep_rate=1.0, dec_lstm_drop=1.0, ann_param=0, optimizer="Adam", lstm_clip_by_norm=5.0, restore=False,
    LOG_DIR="./model_logs/", save_params=0, no_encoder=False, vocab_size=None, coco_dir=_COCO, logging=False,
    hdf5_file=_COCO + "train_val.hdf5", use_hdf5=True, fine_tune=False, fine_tune_top=True, fine_tune_fe=True,
    cnn_lr=0.00001, cnn_optimizer="Adam", cnn_dropout=0.5, weight_decay=0.00004,
    gen_name="00", checkpoint="last_run", num_epochs_per_decay=5, use_c_v=False, gen_val_captions=4000, keep_words=3,
    cap_max_length=100, prior="Normal", max_checkpoints_to_keep=5, mode="training", num_ex_per_epoch=150000,
    image_net_weights_path="./utils/vgg16_weights.npz",
    # additive (not in the reference)
    synthetic=False, seed=1234, max_steps=0, captions_json=None, features_pickle=None, cluster_pickle=None, ckpt_format="tf",
)

# (flag, attribute, converter or "flag" for store_true, choices).  The reference's flags first, in its order.
_FLAGS = [
    ("--lr", "learning_rate", float, None), ("--embed_dim", "embed_size", int, None), ("--enc_hid", "encoder_hidden", int, None),
    ("--dec_hid", "decoder_hidden", int, None), ("--latent", "latent_size", int, None), ("--restore", "restore", "flag", None),
    ("--gpu", None, str, None), ("--coco_dir", "coco_dir", str, None), ("--epochs", "num_epochs", int, None),
    ("--bs", "batch_size", int, None), ("--no_encoder", "no_encoder", "flag", None), ("--temperature", "temperature", float, None),
    ("--gen_name", "gen_name", str, None), ("--dec_drop", "dec_keep_rate", float, None), ("--gen_z_samples", "gen_z_samples", int, None),
    ("--ann_param", "ann_param", float, None), ("--dec_lstm_drop", "dec_lstm_drop", float, None), ("--sample_gen", "sample_gen", str, None),
    ("--checkpoint", "checkpoint", str, None), ("--optimizer", "optimizer", str, ["SGD", "Adam", "Momentum"]),
    ("--c_v", "use_c_v", "flag", None), ("--std", "std", float, None), ("--save_params", "save_params", "flag", None),
    ("--prior", "prior", str, ["GMM", "AG", "Normal"]), ("--fine_tune", "fine_tune", "flag", None),
    ("--mode", "mode", str, ["training", "inference"]),
    # additive
    ("--synthetic", "synthetic", "flag", None), ("--vocab", None, int, None), ("--seed", "seed", int, None),
    ("--max_steps", "max_steps", int, None), ("--captions_json", "captions_json", str, None),
    ("--features_pickle", "features_pickle", str, None), ("--cluster_pickle", "cluster_pickle", str, None),
    ("--ckpt_format", "ckpt_format", str, ["tf", "npz"]),
]
_HELP = {"--synthetic": "train on seeded synthetic batches (no MSCOCO needed)", "--vocab": "vocabulary size for --synthetic (default 10000)",
         "--max_steps": "steps per epoch (0 = the reference's num_ex_per_epoch rule, main.py:217-221)",
         "--captions_json": "COCO captions json (with --features_pickle: in-memory real-data path)",
         "--features_pickle": "pickle {file_name: fc2 feature [1, 4096]} (the reference's ./pickles/<split>.pickle format)",
         "--cluster_pickle": "pickle {file_name: 91-vector} (the reference's ./obj_vectors/c_v.pickle)",
         "--ckpt_format": "tf = TensorFlow V2 checkpoint files (what tf.train.Saver writes), npz = numpy archive"}


class Parameters(object):
    """Attributes = the keys of _DEFAULTS (class-level, so `Parameters().x` and pickled instances behave like the reference's)."""

    def build_parser(self):
        ap = argparse.ArgumentParser(description="CVAE / AG-CVAE captioning trainer (MI355X)")
        for flag, attr, conv, choices in _FLAGS:
            kw = dict(help=_HELP.get(flag))
            if conv == "flag":
                ap.add_argument(flag, action="store_true", **kw)
            else:
                ap.add_argument(flag, default=None, choices=choices, **kw)
        return ap

    def parse_args(self, argv=None):
        args = vars(self.build_parser().parse_args(argv))
        for flag, attr, conv, _ in _FLAGS:
            val = args[flag.lstrip("-")]
            if attr is None:
                continue
            if conv == "flag":
                if val:  # store_true flags only ever switch something on
                    setattr(self, attr, True if attr != "save_params" else 1)
                elif attr in ("restore", "no_encoder", "use_c_v", "fine_tune", "synthetic"):
                    setattr(self, attr, False)
            elif val is not None:
                setattr(self, attr, conv(val))
            else:
                setattr(self, attr, getattr(self, attr))  # materialise the default on the instance (it is pickled by --save_params)
        if self.synthetic:
            self.vocab_size = int(args["vocab"]) if args["vocab"] is not None else 10000
        self.hdf5_file = self.coco_dir + os.path.basename(self.hdf5_file)  # the image array lives next to the data set
        if args["gpu"] is not None:
            os.environ["HIP_VISIBLE_DEVICES"] = str(args["gpu"])
        return self


for _name, _value in _DEFAULTS.items():
    setattr(Parameters, _name, _value)

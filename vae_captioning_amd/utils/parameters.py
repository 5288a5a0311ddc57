"""Run configuration: attribute names, defaults and command-line flags of the
reference's ``Parameters`` (utils/parameters.py:2-66 defaults, :75-132 flags) so
that ``main.py`` invocations keep working unchanged.  New flags are additive.

Differences (documented):
  * ``--gpu`` maps to HIP_VISIBLE_DEVICES (comma list allowed for data-parallel
    runs) and is optional: without it every visible device is used.
"""
import argparse
import os


class Parameters(object):
    # -- general (utils/parameters.py:2-9)
    latent_size = 150
    num_clusters = 90
    num_epochs = 20
    learning_rate = 0.0005
    num_captions = 5
    batch_size = 32
    cnn_feature_size = 4096
    # -- decoding (:10-18)
    temperature = 1.0
    sample_gen = "beam_search"
    beam_size = 10
    # -- encoder / decoder (:19-33)
    encoder_rnn_layers = 1
    encoder_hidden = 512
    std = 0.1
    decoder_hidden = 512
    decoder_rnn_layers = 1
    dec_keep_rate = 1.0
    embed_size = 256
    gen_max_len = 30
    gen_z_samples = 100
    ann_param = 0
    dec_lstm_drop = 1.0
    optimizer = "Adam"
    lstm_clip_by_norm = 5.0
    restore = False
    # -- technical (:36-41)
    LOG_DIR = "./model_logs/"
    save_params = 0
    no_encoder = False
    vocab_size = None
    coco_dir = "/home/luoyy16/datasets-large/mscoco/coco/"
    # -- fine-tuning (:42-51)
    hdf5_file = coco_dir + "train_val.hdf5"
    use_hdf5 = True
    fine_tune = False
    fine_tune_top = True
    fine_tune_fe = True
    cnn_lr = 0.00001
    cnn_optimizer = "Adam"
    cnn_dropout = 0.5
    weight_decay = 0.00004
    # -- inference / preprocessing (:52-66)
    gen_name = "00"
    checkpoint = "last_run"
    num_epochs_per_decay = 5
    use_c_v = False
    gen_val_captions = 4000
    keep_words = 3
    cap_max_length = 100
    prior = "Normal"
    max_checkpoints_to_keep = 5
    mode = "training"
    num_ex_per_epoch = 150000
    image_net_weights_path = "./utils/vgg16_weights.npz"
    logging = False
    # -- additive (not in the reference)
    synthetic = False     # train on seeded synthetic batches (no MSCOCO needed)
    seed = 1234
    max_steps = 0         # 0 = reference stop rule (main.py:217-221)
    captions_json = None
    features_pickle = None
    cluster_pickle = None
    ckpt_format = "tf"    # "tf": TensorFlow V2 checkpoint files (what tf.train.Saver writes); "npz": numpy archive

    def build_parser(self):
        p = argparse.ArgumentParser(description="CVAE / AG-CVAE captioning trainer (MI355X)")
        a = p.add_argument
        a("--lr", default=self.learning_rate, dest="lr")
        a("--embed_dim", default=self.embed_size, dest="embed")
        a("--enc_hid", default=self.encoder_hidden, dest="enc_hid")
        a("--dec_hid", default=self.decoder_hidden, dest="dec_hid")
        a("--latent", default=self.latent_size, dest="latent")
        a("--restore", action="store_true")
        a("--gpu", default=None)
        a("--coco_dir", default=self.coco_dir)
        a("--epochs", default=self.num_epochs)
        a("--bs", default=self.batch_size)
        a("--no_encoder", action="store_true")
        a("--temperature", default=self.temperature)
        a("--gen_name", default=self.gen_name)
        a("--dec_drop", default=self.dec_keep_rate)
        a("--gen_z_samples", default=self.gen_z_samples)
        a("--ann_param", default=self.ann_param)
        a("--dec_lstm_drop", default=self.dec_lstm_drop)
        a("--sample_gen", default=self.sample_gen)
        a("--checkpoint", default=self.checkpoint)
        a("--optimizer", default=self.optimizer, choices=["SGD", "Adam", "Momentum"])
        a("--c_v", default=False, action="store_true")
        a("--std", default=self.std)
        a("--save_params", action="store_true")
        a("--prior", default=self.prior, choices=["GMM", "AG", "Normal"])
        a("--fine_tune", action="store_true")
        a("--mode", default=self.mode, choices=["training", "inference"])
        # additive
        a("--synthetic", action="store_true")
        a("--vocab", default=10000, help="vocabulary size for --synthetic")
        a("--seed", default=self.seed)
        a("--max_steps", default=self.max_steps)
        a("--captions_json", default=None, help="COCO captions json (real-data path; with --features_pickle)")
        a("--features_pickle", default=None, help="pickle {file_name: fc2 feature [1,4096]} (reference format)")
        a("--ckpt_format", default=self.ckpt_format, choices=["tf", "npz"], help="checkpoint file format")
        a("--cluster_pickle", default=None, help="pickle {file_name: 91-vector} (reference ./obj_vectors/c_v.pickle)")
        return p

    def parse_args(self, argv=None):
        args = self.build_parser().parse_args(argv)
        self.learning_rate = float(args.lr)
        self.embed_size = int(args.embed)
        self.encoder_hidden = int(args.enc_hid)
        self.decoder_hidden = int(args.dec_hid)
        self.latent_size = int(args.latent)
        self.restore = args.restore
        self.coco_dir = args.coco_dir
        self.num_epochs = int(args.epochs)
        self.no_encoder = args.no_encoder
        self.temperature = float(args.temperature)
        self.gen_name = args.gen_name
        self.dec_keep_rate = float(args.dec_drop)
        self.gen_z_samples = int(args.gen_z_samples)
        self.ann_param = float(args.ann_param)
        self.dec_lstm_drop = float(args.dec_lstm_drop)
        self.sample_gen = args.sample_gen
        self.checkpoint = args.checkpoint
        self.optimizer = args.optimizer
        self.use_c_v = args.c_v
        self.batch_size = int(args.bs)
        self.std = float(args.std)
        self.save_params = args.save_params
        self.prior = args.prior
        self.fine_tune = args.fine_tune
        self.mode = args.mode
        self.synthetic = args.synthetic
        self.seed = int(args.seed)
        self.max_steps = int(args.max_steps)
        self.captions_json, self.features_pickle, self.cluster_pickle = args.captions_json, args.features_pickle, args.cluster_pickle
        self.ckpt_format = args.ckpt_format
        if self.synthetic:
            self.vocab_size = int(args.vocab)
        self.hdf5_file = self.coco_dir + self.hdf5_file.split("/")[-1]
        if args.gpu is not None:
            os.environ["HIP_VISIBLE_DEVICES"] = str(args.gpu)
        return self

#!/usr/bin/env python
"""Training / inference driver with the command line of the reference's main.py
(utils/parameters.py:75-132) on the MI355X-native hot path.

    python main.py --gpu 0 --synthetic [--prior AG --c_v] [--fine_tune] [--mode inference]
    python -m torch.distributed.run --nproc-per-node 8 main.py --synthetic --fine_tune   # data parallel

The flow follows main.py:43-297 step for step, with the graph-building classes replaced by the
eager facades of vae_captioning_amd (same names, same call order):
    vgg16 -> imf_emb/cv_emb -> Encoder.q_net -> KL -> Decoder.px_z_fi -> masked CE ->
    non_cnn_optimizer / cnn_optimizer -> print every 500 steps -> validate -> checkpoint per epoch.
The MSCOCO data layer (utils/data.py, batch_gen.py) is out of scope; `--synthetic` feeds seeded
batches with the same tensor contract (vae_captioning_amd/synth.py).
"""
import json
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for multi-process GPU work (RCCL peer mappings)
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vae_captioning_amd import session, spec, synth  # noqa: E402
from vae_captioning_amd.ops import optimizers  # noqa: E402
from vae_captioning_amd.utils.parameters import Parameters  # noqa: E402
from vae_captioning_amd.vae_model.decoder import Decoder  # noqa: E402
from vae_captioning_amd.vae_model.encoder import Encoder  # noqa: E402


class SyntheticDictionary(object):
    """Stand-in for utils/captions.py's Dictionary: ids 0 = <PAD>, 1 = <BOS>, 2 = <EOS>."""

    def __init__(self, vocab_size):
        self.vocab_size = vocab_size
        self.idx2word = {0: "<PAD>", synth.BOS: "<BOS>", synth.EOS: "<EOS>"}
        for i in range(3, vocab_size):
            self.idx2word[i] = "w%d" % i
        self.word2idx = {w: i for i, w in self.idx2word.items()}


class SyntheticBatches(object):
    """next_batch() yields (images_or_features, (inputs, labels), lengths, c_v) already flattened to
    N = B * num_captions rows -- the state after preprocess_captions (main.py:226-228)."""

    def __init__(self, params, steps, seed, T=20):
        self.p, self.steps, self.T = params, steps, T
        self.rng = np.random.default_rng(seed)

    def next_batch(self):
        p = self.p
        for _ in range(self.steps):
            b = synth.make_batch(self.rng, p.batch_size, p.num_captions, self.T, p.vocab_size, use_ci=spec.uses_ci(p),
                                 images=p.fine_tune, variable_len=True)
            yield b


def coco_batches(gen, params, epoch_rule=False, world=1):
    """Batch_Generator.next_batch items -> Trainer.set_batch dicts (main.py:222-238).  epoch_rule: keep
    re-shuffling and passing over the data until steps * batch_size > num_ex_per_epoch (main.py:217-221,256-259;
    --max_steps overrides); batch_size is the GLOBAL batch (world * per-rank batch) in data-parallel runs."""
    from vae_captioning_amd.utils.batch_gen import feed_dict
    steps = 0
    while True:
        for images, captions, lengths, c_v in gen.next_batch(use_obj_vectors=params.use_c_v, num_captions=params.num_captions):
            yield feed_dict(images, captions, lengths, c_v, params.num_captions, params.fine_tune)
            steps += 1
            if epoch_rule and (steps >= params.max_steps if params.max_steps else steps * params.batch_size * world > params.num_ex_per_epoch):
                return
        if not epoch_rule:
            return


def main(params):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    backend = os.environ.get("VC_DIST_BACKEND", "nccl")  # gloo: several ranks on ONE GPU (tests only)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local if backend == "nccl" else local % torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    real = coco_train = coco_val = coco_test = None
    if params.captions_json and params.features_pickle:
        # precomputed-feature path of the reference's data layer (utils/data.py + utils/batch_gen.py)
        from vae_captioning_amd.utils.batch_gen import BatchGenerator
        from vae_captioning_amd.utils.captions import Captions, Dictionary
        caps = Captions(params.captions_json)
        cap_dict = Dictionary(caps.captions, params.keep_words)
        os.makedirs("./pickles", exist_ok=True)  # utils/captions.py:122-125: the vocabulary source gen_caption.py reloads
        with open("./pickles/capt_vocab.pickle", "wb") as wf:
            pickle.dump(file=wf, obj=caps.captions)
        with open(params.features_pickle, "rb") as rf:
            feats = pickle.load(rf)
        cvs = None
        if params.cluster_pickle:
            with open(params.cluster_pickle, "rb") as rf:
                cvs = pickle.load(rf)
        real = BatchGenerator(caps.index_captions(cap_dict.word2idx), feats, params.batch_size, cvs, seed=params.seed,
                              shard=(rank, world) if world > 1 else None)
    elif params.synthetic:
        cap_dict = SyntheticDictionary(params.vocab_size)
    else:
        # the reference's own path (main.py:23-41): MSCOCO directory -> Data -> train / val / test generators
        if not os.path.exists(params.coco_dir + "annotations/captions_train2014.json"):
            raise SystemExit("no MSCOCO under --coco_dir %r: give --synthetic, or --captions_json + --features_pickle" % params.coco_dir)
        from vae_captioning_amd.utils.data import Data
        repartiton = params.gen_val_captions >= 0  # main.py:20-26: train on train + val minus the last gen_val_captions images
        coco = Data(params, extract_features=not params.fine_tune, weights_path=params.image_net_weights_path,
                    repartiton=repartiton and not params.fine_tune, gen_val_cap=params.gen_val_captions)
        cap_dict = coco.dictionary
        coco_train = coco.load_train_data_generator(params.batch_size, params.fine_tune, shard=(rank, world) if world > 1 else None)
        coco_val = coco.get_valid_data(params.batch_size, val_tr_unused=coco_train.unused_cap_in, pretrained=not params.fine_tune)
        coco_test = coco.get_test_data(params.batch_size, pretrained=not params.fine_tune) if os.path.exists(coco.test_cap_json) else None
        os.makedirs("./pickles", exist_ok=True)
        with open("./pickles/capt_vocab.pickle", "wb") as wf:  # utils/captions.py:122-125
            pickle.dump(file=wf, obj=coco.captions_tr.captions)
    params.vocab_size = cap_dict.vocab_size  # main.py:92
    from vae_captioning_amd.trainer import Trainer
    tr = Trainer(params, params.vocab_size, world=world, rank=rank, seed=params.seed)
    params._vc_trainer = tr  # the facades below share it (session.get)
    tr.load_state_dict({**spec.init_caption_params(params, params.vocab_size, seed=params.seed),
                        **(spec.init_vgg_params(seed=params.seed) if params.fine_tune else {})})
    if params.mode == "training" and not params.restore:
        # main.py:205-208: the ImageNet weights are loaded whether or not the CNN is fine-tuned (the VGG16 variables always
        # exist in the reference graph and are saved with every checkpoint, quirk Q22)
        if os.path.exists(params.image_net_weights_path):
            print("Loading imagenet weights for futher usage")
            if params.fine_tune:
                tr.vgg.load_weights(params.image_net_weights_path)
            else:
                from vae_captioning_amd.trainer import imagenet_weights
                tr.cnn_host = imagenet_weights(params.image_net_weights_path)
        elif params.fine_tune:
            print("No %s: VGG16 starts from random weights" % params.image_net_weights_path)
        else:
            print("No %s: checkpoints of this run will not contain the cnn/* variables" % params.image_net_weights_path)
    # saver.save(sess, "./checkpoints/{}.ckpt") (main.py:286-288): TF V2 checkpoint files by default,
    # --ckpt_format npz for a name-keyed numpy archive
    ckpt = "./checkpoints/%s.ckpt" % params.checkpoint + (".npz" if params.ckpt_format == "npz" else "")
    if params.restore or params.mode == "inference":
        have = ckpt if params.ckpt_format == "npz" else ckpt + ".index"
        if not os.path.exists(have):  # saver.restore raises NotFoundError (main.py:209-212, ops/inference.py:6-7)
            raise FileNotFoundError("checkpoint %s not found (--restore / --mode inference need ./checkpoints/%s.ckpt*)" % (have, params.checkpoint))
        print("Restoring from checkpoint")
        tr.restore(ckpt)
    steps_per_epoch = params.max_steps or (params.num_ex_per_epoch // (params.batch_size * world) + 1)  # main.py:217-221, global batch
    say = print if rank == 0 else (lambda *a, **k: None)

    if params.mode == "training":
        cap = tr.cap
        encoder = None if params.no_encoder else Encoder(None, None, None, params)
        decoder = Decoder(None, None, None, params, cap_dict)
        optimize, global_step, global_norm = optimizers.non_cnn_optimizer(None, params)
        optimize_cnn = (lambda: 0.0)
        if params.fine_tune:
            optimize_cnn, _ = optimizers.cnn_optimizer(None, params)
        gs = 0  # host mirror of global_step (it restarts at 0 on --restore, Q11): no device sync on non-print steps
        for e in range(params.num_epochs):
            if real is not None:
                it = real.next_batch(use_obj_vectors=spec.uses_ci(params), num_captions=params.num_captions)
            elif coco_train is not None:
                it = coco_batches(coco_train, params, epoch_rule=True, world=world)
            else:
                it = SyntheticBatches(params, steps_per_epoch, params.seed + 17 * e + rank).next_batch()
            for batch in it:
                if batch["cap_dec"].shape[0] != params.batch_size * params.num_captions:
                    continue  # ragged last batch: shapes are static on device
                tr.set_batch(batch)
                # ---- one sess.run([kld, rec_loss, lower_bound, optimize, optimize_cnn, annealing]) ----
                feats = None
                if tr.vgg is not None:
                    feats = tr.vgg.forward(tr.images, cap.step)
                    if tr.vgg.wd:
                        tr.vgg.reg_sumsq(cap.red.data_ptr() + 12)
                images_fv = cap.fw_prepare(feats)                       # main.py:84-108
                if encoder is not None:
                    encoder.images_fv = decoder.images_fv = images_fv
                    qz, tm_list, tv_list = encoder.q_net()              # main.py:117
                dec_model, x_logits, shpe, _ = decoder.px_z_fi({} if params.no_encoder else {"z": qz})  # main.py:146-150
                cap.fw_loss(train=True)                                 # main.py:152-177
                optimize()                                              # main.py:179
                optimize_cnn()                                          # main.py:183
                gs += 1
                if (gs - 1) % 500 == 0:
                    kl, rl, lb, ann = tr.losses()
                    say("Epoch: {} Iteration: {} VLB: {} Rec Loss: {}".format(e, gs - 1, lb, rl))
                    if not params.no_encoder:
                        say("Annealing coefficient:{} KLD: {}".format(ann, kl))
            kl, rl, lb, ann = tr.losses()
            say("Epoch: {} Iteration: {} VLB: {} Rec Loss: {}".format(e, int(global_step.item()), lb, rl))
            # validate(): rec_loss of the training graph on held-out batches (main.py:262-284)
            val = []
            held_out = coco_batches(coco_val, params) if coco_val is not None else SyntheticBatches(params, 4, params.seed + 99991 + rank).next_batch()
            for batch in held_out:
                if batch["cap_dec"].shape[0] != params.batch_size * params.num_captions:
                    continue
                tr.set_batch(batch)
                val.append(tr.eval_rec_loss())
            say("Validation reconstruction loss: {}".format(np.mean(val)))
            say("-----------------------------------------------")
            if rank == 0:
                os.makedirs("./checkpoints", exist_ok=True)
                tr.save(ckpt)
                say("Model saved in file: %s" % ckpt)
    if params.mode == "inference":
        # ops/inference.py:4-39: captions for the validation images -> ./val_{gen_name}.json
        decoder = Decoder(None, None, None, params, cap_dict)
        if coco_val is not None:  # ops/inference.py:4-56 on the validation / test image sets
            from vae_captioning_amd.ops.inference import inference
            if rank == 0:
                inference(params, decoder, coco_val, coco_test)
            if world > 1:
                dist.destroy_process_group()
            return
        rng = np.random.default_rng(params.seed + 5)
        captions_gen = []
        for it in range(2):
            b = synth.make_batch(rng, params.batch_size, 1, 20, params.vocab_size, use_ci=spec.uses_ci(params), images=params.fine_tune)
            ids = ["synthetic_%06d" % (it * params.batch_size + i) for i in range(params.batch_size)]
            pics = b["images"] if params.fine_tune else b["features"]
            c_v = b.get("c_v")
            if params.sample_gen == "beam_search":
                sent = decoder.beam_search(None, ids, pics, None, c_v, beam_size=params.beam_size)
            else:
                sent, _ = decoder.online_inference(None, ids, pics, None, c_v=c_v)
            captions_gen += sent
        say("Generated {} captions".format(len(captions_gen)))
        if rank == 0:
            with open("./val_{}.json".format(params.gen_name), "w") as wj:
                json.dump(captions_gen, wj)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    params = Parameters()
    params.parse_args()
    if params.save_params:
        os.makedirs("./pickles", exist_ok=True)
        fn = "./pickles/params_{}_{}_{}_{}.pickle".format(params.prior, params.no_encoder, params.checkpoint, params.use_c_v)
        print("Saving params to: ", fn)
        with open(fn, "wb") as wf:
            pickle.dump(file=wf, obj={k: v for k, v in vars(params).items() if not k.startswith("_")})
    main(params)

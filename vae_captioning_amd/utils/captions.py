"""Caption tokeniser and vocabulary with the behaviour of the reference's utils/captions.py
(host-side data preparation).

Rules reproduced (each is observable in the token ids the model is fed):
  * tokenise: lower-case, split on runs of non-word characters (re `\\W+`), drop empty pieces,
    wrap in <BOS> ... <EOS>                                  (utils/captions.py:38-41)
  * the max_length clip tests len() of the annotation DICT, so it never clips (:32-34)
  * vocabulary: count every token (including <BOS>/<EOS>), add one '<UNK>', sort by
    (-count, word), keep count >= keep_words or '<UNK>', ids 1.., <PAD> = 0  (:95-121)
  * unknown words index to <UNK>                             (:43-60)
"""
import json
import re
from collections import Counter, OrderedDict

_SPLIT = re.compile(r"\W+")


def tokenize(caption):
    return ["<BOS>"] + [w for w in _SPLIT.split(caption.lower()) if w] + ["<EOS>"]


class Captions(object):
    """{file_name: [token lists]} from a COCO captions json (or an already parsed dict)."""

    def __init__(self, captions_file, max_length=16):
        self.max_length = max_length
        j = captions_file if isinstance(captions_file, dict) else json.load(open(captions_file))
        names = {img["id"]: img["file_name"] for img in j["images"]}
        self.filename_to_imid = {img["file_name"]: img["id"] for img in j["images"]}
        self.captions = OrderedDict()
        for ann in j["annotations"]:
            self.captions.setdefault(names[ann["image_id"]], []).append(tokenize(ann["caption"]))
        self.captions_indexed = {k: [list(c) for c in v] for k, v in self.captions.items()}
        self.num_captions = len(self.captions)

    def index_captions(self, word2idx):
        unk = word2idx["<UNK>"]
        for name, caps in self.captions.items():
            self.captions_indexed[name] = [[word2idx.get(w, unk) for w in c] for c in caps]
        return self.captions_indexed


class Dictionary(object):
    def __init__(self, caption_dict, keep_words=3):
        words = []
        for caps in caption_dict.values():
            for cap in caps:
                words += [w if w in ("<EOS>", "<BOS>", "<PAD>") else w.lower() for w in cap]
        words.append("<UNK>")
        kept = [w for w, c in sorted(Counter(words).items(), key=lambda wc: (-wc[1], wc[0])) if c >= keep_words or w == "<UNK>"]
        self.word2idx = {w: i for i, w in enumerate(kept, start=1)}
        self.idx2word = {i: w for w, i in self.word2idx.items()}
        self.idx2word[0] = "<PAD>"
        self.word2idx["<PAD>"] = 0

    @property
    def vocab_size(self):
        return len(self.idx2word)

    def __len__(self):
        return len(self.idx2word)

    def seq2dx(self, sentence):
        return [self.word2idx[w] for w in sentence]

"""Weight files of an experiment: every format the reference's tools leave behind.

  weights/lstm_weights.pkl                     dict name -> float32 array       train/weights.py:70-74
  weights/lstm_weights_comp_{bit}.pkl          same layout, k-means DECODED     train/comp.py:70,79 (model.py:74-78 picks it for comp>0)
  weights/comp_{bit}/lstm_weights_comp_dump.pkl  name -> (code, codebook)        train/comp.py:69,80
  weights/comp_{bit}/{name}_code.txt, {name}_codebook.txt                       train/comp.py:72-74 (debug dumps)
  weights/{name}.npy / {name}.txt (lists: {name}{i}.txt)                        train/weights.py:76-87 (verbose dumps)

``load_weights`` tries them in that order for the requested ``comp`` and returns
the plain dict of float arrays that ``LSTM_Model`` consumes; the (code, codebook)
forms are decoded with ``np.take(codebook, code)`` exactly as train/comp.py:70 does.
"""
import os
import pickle

import numpy as np

from . import config as _config

TENSOR_NAMES = ["HMi", "HMf", "HMo", "HMg", "IMi", "IMf", "IMo", "IMg", "bi", "bf", "bo", "bg", "b2", "PM", "UM", "LM"]


def weights_dir(experiment_id):
    return os.path.join(_config.experiment_path, str(experiment_id), "weights")


def decode_codebook(code, codebook):
    """train/comp.py:70 -- ``np.take(codebook, code)`` (codebook is [2^bit, 1] float32)."""
    return np.take(np.asarray(codebook), np.asarray(code))


def load_codes(experiment_id=0, comp=0):
    """The (code uint8, codebook float32) pairs of a k-means compressed model where train/comp.py left them
    (comp_{bit}/lstm_weights_comp_dump.pkl, or the debug text files) and every code fits a byte -- else None.
    DeviceModel keeps these resident and expands the vocabulary blocks from them on the device (jlm_dequant_u8)."""
    if not comp:
        return None
    cdir = os.path.join(weights_dir(experiment_id), "comp_{}".format(comp))
    dump = os.path.join(cdir, "lstm_weights_comp_dump.pkl")
    pairs = None
    if os.path.exists(dump):
        with open(dump, "rb") as f:
            pairs = pickle.load(f)
    elif os.path.isdir(cdir):
        pairs = {}
        for fn in sorted(os.listdir(cdir)):
            if fn.endswith("_code.txt"):
                name = fn[:-len("_code.txt")]
                pairs[name] = (np.loadtxt(os.path.join(cdir, fn), dtype=np.int64),
                               np.loadtxt(os.path.join(cdir, name + "_codebook.txt"), dtype=np.float32))
    if not pairs:
        return None
    out = {}
    for k, (code, book) in pairs.items():
        code, book = np.asarray(code), np.asarray(book, dtype=np.float32).reshape(-1)
        if book.size > 256 or code.size == 0 or int(code.max()) >= book.size or int(code.min()) < 0:
            continue
        out[k] = (np.ascontiguousarray(code.astype(np.uint8)), book)
    return out or None


def _load_text_tensors(d, config):
    out = {}
    names = list(TENSOR_NAMES)
    if config is not None and config.get("V_table"):
        names.remove("LM")
        for i in range(len(config["embedding_seg"])):
            names.append("LM{}".format(i))
            if i:
                names.append("VT{}".format(i))
    for n in names:
        npy, txt = os.path.join(d, n + ".npy"), os.path.join(d, n + ".txt")
        if os.path.exists(npy):
            out[n] = np.load(npy)
        elif os.path.exists(txt):
            out[n] = np.loadtxt(txt, dtype=np.float32)
        elif n == "LM" and os.path.exists(os.path.join(d, "LM0.txt")):       # D_softmax: list of blocks
            blocks, i = [], 0
            while os.path.exists(os.path.join(d, "LM{}.txt".format(i))):
                blocks.append(np.loadtxt(os.path.join(d, "LM{}.txt".format(i)), dtype=np.float32, ndmin=2))
                i += 1
            out[n] = blocks
    return out


def load_weights(experiment_id=0, comp=0, config=None):
    d = weights_dir(experiment_id)
    if comp:
        pkl = os.path.join(d, "lstm_weights_comp_{}.pkl".format(comp))
        if os.path.exists(pkl):
            print('use compressed model, comp_{}'.format(comp))
            with open(pkl, "rb") as f:
                return pickle.load(f)
        cdir = os.path.join(d, "comp_{}".format(comp))
        dump = os.path.join(cdir, "lstm_weights_comp_dump.pkl")
        if os.path.exists(dump):
            with open(dump, "rb") as f:
                return {k: decode_codebook(code, book) for k, (code, book) in pickle.load(f).items()}
        out = {}
        if os.path.isdir(cdir):
            for fn in sorted(os.listdir(cdir)):
                if fn.endswith("_code.txt"):
                    name = fn[:-len("_code.txt")]
                    code = np.loadtxt(os.path.join(cdir, fn), dtype=np.int64)
                    book = np.loadtxt(os.path.join(cdir, name + "_codebook.txt"), dtype=np.float32)
                    out[name] = decode_codebook(code, book)
        if out:
            return out
        raise FileNotFoundError("no compressed weights for comp={} under {}".format(comp, d))
    pkl = os.path.join(d, "lstm_weights.pkl")
    if os.path.exists(pkl):
        with open(pkl, "rb") as f:
            return pickle.load(f)
    out = _load_text_tensors(d, config)
    if out:
        return out
    raise FileNotFoundError(pkl)

"""Evaluation harness: counterpart of the reference's decoder/eval.py.

Same flags (eval.py:17-28, including the ``type=bool`` behaviour: any non-empty
string is true), same decoder selection order (eval.py:41-48), same eval-set
loader (eval.py:125-166), same log file name/body and the same printed summary
lines (eval.py:65-122) -- those strings are the file format the golden vectors
pin.  Extras: ``--root`` (artefact directory; the reference freezes it from
``__file__``), ``--batch N`` decodes N sentences per GPU launch sequence instead
of one (identical output, much faster).

    python -m jlm_amd.eval --root /path/to/artifacts -e 1 -es 100 -b 10 [--batch 256]
"""
import argparse
import os
import time

import numpy as np

from . import config as _config
from .data import Vocab

# (long flag, short flag, type, default, what it selects) -- eval.py:17-28
_FLAGS = [
    ("experiment_id", "e", int, 1, "experiment directory under train/experiments/"),
    ("eval_size", "es", int, 100, "sentences taken from data/test.txt"),
    ("use_ngram", "ng", bool, False, "decode with the n-gram baseline instead of the LSTM"),
    ("ngram_order", "o", int, 3, "order of the n-gram model"),
    ("comp", "c", int, 0, "bits of the k-means compressed weights to load (0: uncompressed)"),
    ("vocab_select", "vs", bool, False, "normalise over the lattice's own words only"),
    ("top_sampling", "ts", bool, False, "add the `samples` most frequent words to the selected vocabulary"),
    ("random_sampling", "rs", bool, False, "add `samples` random words to the selected vocabulary"),
    ("samples", "s", int, 0, "number of sampled words"),
    ("beam_size", "b", int, 10, "hypotheses kept per frame"),
    ("dynamic_decoding", "dd", bool, False, "incremental vocabulary selection (DynamicDecoder)"),
]
_LOG_NAME = 'eval/eval_log_{}_e_{}_dynamic_{}_size_{}_b_{}_comp_{}_vocab_sel_{}_samples_{}_top_{}_random_{}.txt'


def build_parser():
    parser = argparse.ArgumentParser(description="conversion accuracy of a decoder on data/test.txt")
    for name, short, typ, default, text in _FLAGS:
        parser.add_argument("--" + name, "-" + short, type=typ, default=default, help=text)
    parser.add_argument("--root", default=None, help="artefact root (data/, train/experiments/)")
    parser.add_argument("--batch", type=int, default=1, help="sentences decoded per batch on the GPU")
    return parser


def _surface(token):
    return token.split('/')[0]


def _reading(token):
    parts = token.split('/')
    return parts[1] if parts[1] != '' else parts[0]


class Evaluator:
    def __init__(self, args):
        self.args = args
        self.config = _config.load_config_dict(args.experiment_id)
        if self.config['char_rnn']:          # eval.py:35-36
            from .data import CharVocab
            self.vocab = CharVocab(self.config['vocab_size'])
        else:
            self.vocab = Vocab(self.config['vocab_size'])
        self.w2i = self.vocab.w2i
        self.decoder = self._make_decoder()
        if hasattr(self.decoder, "perf_timing"):
            self.decoder.perf_timing = True       # the log's per-step times (eval.py:104-121) need the per-frame events

    def _make_decoder(self):
        a = self.args                 # selection order of eval.py:41-48
        if a.use_ngram:
            from .decoder_ngram import NGramDecoder
            return NGramDecoder(experiment_id=a.experiment_id, ngram_order=a.ngram_order)
        if self.config['char_rnn']:
            from .decoder_char import CharRNNDecoder
            return CharRNNDecoder(experiment_id=a.experiment_id, comp=a.comp)
        if a.dynamic_decoding:
            from .decoder_dynamic import DynamicDecoder
            return DynamicDecoder(experiment_id=a.experiment_id, comp=a.comp)
        from .decoder import Decoder
        return Decoder(experiment_id=a.experiment_id, comp=a.comp)

    def log_name(self):
        a = self.args
        kind = "ngram_{}".format(a.ngram_order) if a.use_ngram else "neural"
        return _LOG_NAME.format(kind, a.experiment_id, a.dynamic_decoding, a.eval_size, a.beam_size, a.comp, a.vocab_select,
                                a.samples, a.top_sampling, a.random_sampling)

    def _decode_all(self, inputs):
        a = self.args
        kw = dict(beam_width=a.beam_size, vocab_select=a.vocab_select, samples=a.samples,
                  top_sampling=a.top_sampling, random_sampling=a.random_sampling)
        step = max(1, a.batch)
        if step == 1:                 # sentence at a time, as the reference does
            return [self.decoder.decode(x, **kw) for x in inputs]
        out = []
        for i in range(0, len(inputs), step):
            out.extend(self.decoder.decode_batch(inputs[i:i + step], **kw))
        return out

    def _timing_lines(self):
        a, d = self.args, self.decoder
        if a.use_ngram:               # eval.py:104-107: only the neural decoders log per-step times
            return []
        per_step = d.perf_log_lstm + d.perf_log_softmax
        return ["--- %f seconds lstm per step ---" % (np.mean(d.perf_log_lstm)),
                "--- %f seconds softmax per step ---" % (np.mean(d.perf_log_softmax)),
                "--- %f seconds per sent.---" % (np.sum(per_step) / d.perf_sen)]

    def evaluate(self):
        a = self.args
        hits = {"best hit": 0, "nbest hit": 0, "no hit": 0}
        with open(self.log_name(), 'w', encoding='utf-8') as log:
            inputs, targets = self.load_eval_set()
            t_start = time.time()
            for x, y, nbest in zip(inputs, targets, self._decode_all(inputs)):
                sentences = [''.join(_surface(w) for w in words) for _score, words in nbest]
                verdict = "best hit" if y == sentences[0] else ("nbest hit" if y in sentences else "no hit")
                hits[verdict] += 1
                log.write(verdict + '\n')
                log.write('{}\t{}\n'.format(y, x))
                log.writelines(s + '\n' for s in sentences)
            best, nbest_hits = hits["best hit"], hits["nbest hit"]
            summary = 'best_hit {} nbest_hit{} no_hit {} eval_size {}'.format(best, nbest_hits, a.eval_size - best - nbest_hits,
                                                                             a.eval_size)
            timing = self._timing_lines()
            log.write(summary)
            log.writelines(timing)
            log.write("--- %s seconds ---" % (time.time() - t_start))
            print(summary)
            for ln in timing:
                print(ln)
            if a.dynamic_decoding and not a.use_ngram:
                d = self.decoder
                fix = lambda v: ("%f" % np.mean(v)) if len(v) else "n/a"
                print("--- %s seconds per step for vocab fix.---" % fix(d.perf_log_fix_vocab))
                print("--- %s seconds per step for lattice path fix.---" % fix(d.perf_log_fix_lattice_path_prob))
            print("--- %s seconds ---" % (time.time() - t_start))
        return best, nbest_hits

    def load_eval_set(self):
        """The first ``eval_size`` lines of data/test.txt whose tokens are all in-vocabulary (eval.py:125-166):
        x = the tokens' readings (the surface where a token has none), y = their surfaces, both concatenated."""
        want = self.args.eval_size
        xs, ys = [], []
        with open(os.path.join(_config.data_path, 'test.txt'), 'r', encoding='utf-8') as f:
            lines = f.readlines()
        print('take {} for evaluation from all {} lines'.format(want, len(lines)))
        for line in lines:
            tokens = line.strip().split(' ')
            if any(self.decoder._check_oov(t) for t in tokens):
                continue
            xs.append(''.join(_reading(t) for t in tokens))
            ys.append(''.join(_surface(t) for t in tokens))
            if len(xs) >= want:
                break
        print('{} pairs load'.format(len(xs)))
        return xs, ys


def parse_log(top='./'):
    """Print the summary line of every eval log under ``top`` (eval.py:168-178)."""
    for folder, _subs, files in os.walk(top):
        for filename in files:
            if 'eval_log' not in filename:
                continue
            print(filename)
            with open(os.path.join(folder, filename), 'r', encoding='utf-8') as f:
                for line in f:
                    if 'best_hit' in line:
                        print(line.strip())


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.root:
        _config.set_root(args.root)
    os.makedirs('eval', exist_ok=True)
    return Evaluator(args).evaluate()


if __name__ == '__main__':
    main()

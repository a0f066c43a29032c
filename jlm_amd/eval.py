"""Evaluation harness: counterpart of the reference's decoder/eval.py.

Same flags (eval.py:17-28, including the ``type=bool`` behaviour: any non-empty
string is true), same decoder selection order (eval.py:41-48), same eval-set
loader (eval.py:125-166), same log file name/body and the same printed summary
lines (eval.py:65-122).  Extras: ``--root`` (artefact directory; the reference
freezes it from ``__file__``), ``--batch N`` decodes N sentences per GPU launch
sequence instead of one (identical output, much faster).

    python -m jlm_amd.eval --root /path/to/artifacts -e 1 -es 100 -b 10 [--batch 256]
"""
import argparse
import os
import time

import numpy as np

from . import config as _config
from .data import Vocab


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--experiment_id", "-e", type=int, default=1, help="experiment id to eval")
    parser.add_argument("--eval_size", "-es", type=int, default=100, help="Number of sentences to evaluate")
    parser.add_argument("--use_ngram", "-ng", type=bool, default=False, help="Use ngram decoder or not")
    parser.add_argument("--ngram_order", "-o", type=int, default=3, help="Ngram order")
    parser.add_argument("--comp", "-c", type=int, default=0, help="Compression bit, 0 means no compression")
    parser.add_argument("--vocab_select", "-vs", type=bool, default=False, help="Use vocab select method or not")
    parser.add_argument("--top_sampling", "-ts", type=bool, default=False, help="Sampling strategy for vocab select")
    parser.add_argument("--random_sampling", "-rs", type=bool, default=False, help="Sampling strategy for vocab select")
    parser.add_argument("--samples", "-s", type=int, default=0, help="Samples when using advanced sampling")
    parser.add_argument("--beam_size", "-b", type=int, default=10, help="Beam size for decoder")
    parser.add_argument("--dynamic_decoding", "-dd", type=bool, default=False, help="Use incremental decoding or not")
    parser.add_argument("--root", default=None, help="artefact root (data/, train/experiments/)")
    parser.add_argument("--batch", type=int, default=1, help="sentences decoded per batch on the GPU")
    return parser


class Evaluator:
    def __init__(self, args):
        self.args = args
        self.config = _config.load_config_dict(args.experiment_id)
        if self.config['char_rnn']:
            raise NotImplementedError("char-RNN models are outside the scope of this build (SURVEY.md 8f)")
        self.vocab = Vocab(self.config['vocab_size'])
        self.w2i = self.vocab.w2i
        if args.use_ngram:            # the CPU baseline (eval.py:41-42)
            from .decoder_ngram import NGramDecoder
            self.decoder = NGramDecoder(experiment_id=args.experiment_id, ngram_order=args.ngram_order)
        elif args.dynamic_decoding:
            from .decoder_dynamic import DynamicDecoder
            self.decoder = DynamicDecoder(experiment_id=args.experiment_id, comp=args.comp)
        else:
            from .decoder import Decoder
            self.decoder = Decoder(experiment_id=args.experiment_id, comp=args.comp)

    def log_name(self):
        a = self.args
        return 'eval/eval_log_{}_e_{}_dynamic_{}_size_{}_b_{}_comp_{}_vocab_sel_{}_samples_{}_top_{}_random_{}.txt'.format(
            "ngram_{}".format(a.ngram_order) if a.use_ngram else "neural", a.experiment_id, a.dynamic_decoding, a.eval_size, a.beam_size, a.comp, a.vocab_select, a.samples,
            a.top_sampling, a.random_sampling)

    def evaluate(self):
        a = self.args
        best_hit = 0
        n_best_hit = 0
        with open(self.log_name(), 'w', encoding='utf-8') as f:
            x_, y_ = self.load_eval_set()
            start_time = time.time()
            kw = dict(beam_width=a.beam_size, vocab_select=a.vocab_select, samples=a.samples,
                      top_sampling=a.top_sampling, random_sampling=a.random_sampling)
            all_results = []
            step = max(1, a.batch)
            for i in range(0, len(x_), step):
                if step == 1:
                    all_results.append(self.decoder.decode(x_[i], **kw))
                else:
                    all_results.extend(self.decoder.decode_batch(x_[i:i + step], **kw))
            for x, y, results in zip(x_, y_, all_results):
                sentences = [''.join([w.split('/')[0] for w in item[1]]) for item in results]
                if y == sentences[0]:
                    best_hit += 1
                    f.write('best hit\n')
                elif y in sentences:
                    f.write('nbest hit\n')
                    n_best_hit += 1
                else:
                    f.write('no hit\n')
                f.write('{}\t{}\n'.format(y, x))
                for item in sentences:
                    f.write('{}\n'.format(item))
            summary = 'best_hit {} nbest_hit{} no_hit {} eval_size {}'.format(
                best_hit, n_best_hit, a.eval_size - best_hit - n_best_hit, a.eval_size)
            f.write(summary)
            d = self.decoder
            lines = []
            if not a.use_ngram:       # eval.py:104-107
                lines = ["--- %f seconds lstm per step ---" % (np.mean(d.perf_log_lstm)),
                         "--- %f seconds softmax per step ---" % (np.mean(d.perf_log_softmax)),
                         "--- %f seconds per sent.---" % (np.sum(d.perf_log_lstm + d.perf_log_softmax) / d.perf_sen)]
            for ln in lines:
                f.write(ln)
            f.write("--- %s seconds ---" % (time.time() - start_time))
            print(summary)
            for ln in lines:
                print(ln)
            if a.dynamic_decoding and not a.use_ngram:
                print("--- %f seconds per step for vocab fix.---" % np.mean(d.perf_log_fix_vocab))
                print("--- %f seconds per step for lattice path fix.---" % np.mean(d.perf_log_fix_lattice_path_prob))
            print("--- %s seconds ---" % (time.time() - start_time))
        return best_hit, n_best_hit

    def load_eval_set(self):
        """reference eval.py:125-166: first eval_size lines of data/test.txt whose
        tokens are all in-vocabulary; x = readings, y = display strings."""
        a = self.args
        x, y = [], []
        with open(os.path.join(_config.data_path, 'test.txt'), 'r', encoding='utf-8') as f:
            lines = f.readlines()
            print('take {} for evaluation from all {} lines'.format(a.eval_size, len(lines)))
            for line in lines:
                tokens = line.strip().split(' ')
                if any(self.decoder._check_oov(t) for t in tokens):
                    continue
                readings = ''.join([t.split('/')[1] if t.split('/')[1] != '' else t.split('/')[0] for t in tokens])
                target = ''.join([t.split('/')[0] for t in tokens])
                x.append(readings)
                y.append(target)
                if len(x) >= a.eval_size:
                    break
            print('{} pairs load'.format(len(x)))
        return x, y


def parse_log():
    """reference eval.py:168-178."""
    for folder, _subs, files in os.walk('./'):
        for filename in files:
            if 'eval_log' in filename:
                print(filename)
                with open(os.path.join(folder, filename), 'r', encoding='utf-8') as f:
                    for line in f.readlines():
                        if 'best_hit' in line:
                            print(line.strip())


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.root:
        _config.set_root(args.root)
    os.makedirs('eval', exist_ok=True)
    ev = Evaluator(args)
    return ev.evaluate()


if __name__ == '__main__':
    main()

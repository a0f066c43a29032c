"""Run the reference's decoder/eval.py UNCHANGED on the MI355X implementation.

    JLM_ROOT=/artifacts python tools/run_reference_eval.py /path/to/JLM/decoder/eval.py -e 1 -es 100 -b 10

Must be started in a directory that has an `eval/` sub-directory (the reference
writes its log there, decoder/eval.py:65)."""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(eval_py, argv):
    for p in (REPO, os.path.join(REPO, "compat")):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "compat"))
    for m in ("config", "model", "decoder", "decoder_dynamic", "decoder_ngram", "train", "train.data"):
        sys.modules.pop(m, None)
    old = sys.argv
    sys.argv = [eval_py] + list(argv)
    try:
        runpy.run_path(eval_py, run_name="__main__")
    finally:
        sys.argv = old


if __name__ == "__main__":
    os.makedirs("eval", exist_ok=True)
    run(sys.argv[1], sys.argv[2:])

"""reference decoder/decoder_ngram.py: the n-gram baseline is out of scope
(SURVEY.md 8f); the name exists so that eval.py's import resolves."""


class NGramDecoder:
    def __init__(self, *a, **k):
        raise NotImplementedError("NGramDecoder is outside the scope of this build (SURVEY.md 8f)")

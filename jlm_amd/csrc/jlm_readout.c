/* jlm_readout.c -- CPython extension jlm_amd._readout: the n-best lists of a batch as Python objects.
 *
 * Decoder.decode returns [(neg_log_prob, [word, ...])] (decoder/decoder.py:236-241: the words of every
 * surviving path, `<eos>` dropped, best first, at most topN).  The device leaves the back-traces of all paths
 * (jlm_backtrace, include/jlm_hip.h); turning them into 2 560 tuples and lists per batch in Python costs as
 * much as the GPU needs for the whole batch, so this one call builds them with the C API.  Host logic only;
 * jlm_amd/engine.py keeps the equivalent numpy implementation (DecodeEngine._read_out_py) and the tests
 * compare the two.
 *
 * The containers built here hold strings, floats and each other -- they cannot be part of a reference cycle unless the caller
 * later makes one through them -- and are taken out of the cyclic collector's lists (PyObject_GC_UnTrack, as CPython itself
 * does for tuples and dicts of atomic contents): 8 k containers per 256-sentence batch otherwise make every collection of the
 * caller's process scan the results decoded so far (2 ms per 40-batch call when the collector catches up; quadratic with the
 * collector left on during the call).  Reference counting frees them as usual.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <stdlib.h>

/* JLM_TRACK_RESULTS=1: leave the containers in the collector's lists (a host that links its own objects into the
 * returned lists and relies on cycle collection through them). */
static int jlm_untrack = -1;
#define JLM_UNTRACK(o) do { if (jlm_untrack) PyObject_GC_UnTrack(o); } while (0)

typedef struct { Py_buffer b; int ok; } Buf;

static int get(PyObject *o, Buf *x, Py_ssize_t itemsize, const char *what) {
    x->ok = 0;
    if (PyObject_GetBuffer(o, &x->b, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return -1;
    x->ok = 1;
    if (x->b.itemsize != itemsize) {
        PyErr_Format(PyExc_TypeError, "%s: item size %zd, expected %zd", what, x->b.itemsize, itemsize);
        return -1;
    }
    return 0;
}

/* nbest(nodes, lens, scores, node_lex, node_sent, node_start, lex_words, texts, n_sent, beam, top, stride) */
static PyObject *nbest(PyObject *self, PyObject *args) {
    PyObject *o_nodes, *o_len, *o_score, *o_lex, *o_sent, *o_start, *lex_words, *texts;
    int B, beam, top, stride;
    if (!PyArg_ParseTuple(args, "OOOOOOO!O!iiii", &o_nodes, &o_len, &o_score, &o_lex, &o_sent, &o_start, &PyList_Type,
                          &lex_words, &PyList_Type, &texts, &B, &beam, &top, &stride))
        return NULL;
    Buf nodes, lens, score, lex, sent, start;
    nodes.ok = lens.ok = score.ok = lex.ok = sent.ok = start.ok = 0;
    PyObject *out = NULL, *eos = NULL;
    if (get(o_nodes, &nodes, 4, "nodes") || get(o_len, &lens, 4, "lens") || get(o_score, &score, 8, "scores") ||
        get(o_lex, &lex, 4, "node_lex") || get(o_sent, &sent, 4, "node_sent") || get(o_start, &start, 4, "node_start"))
        goto done;
    {
        const Py_ssize_t rmax = (Py_ssize_t)B * beam;
        const Py_ssize_t n_nodes = lex.b.len / 4, n_lex = PyList_GET_SIZE(lex_words), n_text = PyList_GET_SIZE(texts);
        if (B < 0 || beam <= 0 || stride <= 0 || nodes.b.len / 4 < rmax * stride || lens.b.len / 4 < rmax ||
            score.b.len / 8 < rmax || sent.b.len / 4 < n_nodes || start.b.len / 4 < n_nodes) {
            PyErr_SetString(PyExc_ValueError, "nbest: array shapes do not cover n_sent * beam paths");
            goto done;
        }
        const int32_t *pn = (const int32_t *)nodes.b.buf, *pl = (const int32_t *)lens.b.buf;
        const int32_t *plex = (const int32_t *)lex.b.buf, *psent = (const int32_t *)sent.b.buf;
        const int32_t *pstart = (const int32_t *)start.b.buf;
        const double *ps = (const double *)score.b.buf;
        const int R = top < beam ? top : beam;
        if (jlm_untrack < 0) { const char *e = getenv("JLM_TRACK_RESULTS"); jlm_untrack = !(e && e[0] == '1'); }
        eos = PyUnicode_FromString("<eos>");
        out = eos ? PyList_New(B) : NULL;
        if (!out) goto done;
        for (Py_ssize_t s = 0; s < B; ++s) {
            int nr = 0;                    /* ranks are filled from 0; the list ends at the first empty one */
            while (nr < R && pl[s * beam + nr] > 0) ++nr;
            PyObject *lst = PyList_New(nr);
            if (!lst) goto fail;
            PyList_SET_ITEM(out, s, lst);
            JLM_UNTRACK(lst);
            for (int r = 0; r < nr; ++r) {
                const Py_ssize_t row = s * beam + r;
                const int k = pl[row] - 1;             /* the trace ends at the root (<eos>), which is dropped */
                if (k >= stride) { PyErr_SetString(PyExc_ValueError, "nbest: trace longer than stride"); goto fail; }
                PyObject *words = PyList_New(k);
                if (!words) goto fail;
                JLM_UNTRACK(words);
                for (int j = 0; j < k; ++j) {          /* traces run from the last word back */
                    const int32_t id = pn[row * stride + (k - 1 - j)];
                    PyObject *w;
                    if (id < 0 || id >= n_nodes) { Py_DECREF(words); PyErr_SetString(PyExc_IndexError, "nbest: node id"); goto fail; }
                    const int32_t lx = plex[id];
                    if (lx >= 0) {
                        if (lx >= n_lex) { Py_DECREF(words); PyErr_SetString(PyExc_IndexError, "nbest: lexicon index"); goto fail; }
                        w = PyList_GET_ITEM(lex_words, lx);
                        Py_INCREF(w);
                    } else if (lx == -1) {
                        w = eos;
                        Py_INCREF(w);
                    } else {                           /* <unk> fallback node: the raw kana (decoder.py:128-130) */
                        const int32_t si = psent[id];
                        if (si < 0 || si >= n_text) { Py_DECREF(words); PyErr_SetString(PyExc_IndexError, "nbest: sentence index"); goto fail; }
                        w = PySequence_GetItem(PyList_GET_ITEM(texts, si), pstart[id]);
                        if (!w) { Py_DECREF(words); goto fail; }
                    }
                    PyList_SET_ITEM(words, j, w);
                }
                PyObject *sc = PyFloat_FromDouble(ps[row]);
                PyObject *tup = sc ? PyTuple_New(2) : NULL;
                if (!tup) { Py_XDECREF(sc); Py_DECREF(words); goto fail; }
                JLM_UNTRACK(tup);
                PyTuple_SET_ITEM(tup, 0, sc);
                PyTuple_SET_ITEM(tup, 1, words);
                PyList_SET_ITEM(lst, r, tup);
            }
        }
        goto done;
    }
fail:
    Py_CLEAR(out);
done:
    Py_XDECREF(eos);
    if (nodes.ok) PyBuffer_Release(&nodes.b);
    if (lens.ok) PyBuffer_Release(&lens.b);
    if (score.ok) PyBuffer_Release(&score.b);
    if (lex.ok) PyBuffer_Release(&lex.b);
    if (sent.ok) PyBuffer_Release(&sent.b);
    if (start.ok) PyBuffer_Release(&start.b);
    return out;
}

static PyMethodDef methods[] = {
    {"nbest", nbest, METH_VARARGS,
     "nbest(nodes, lens, scores, node_lex, node_sent, node_start, lex_words, texts, n_sent, beam, top, stride)\n"
     "-> [[(score, [word, ...]), ...] per sentence] from jlm_backtrace's arrays."},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_readout", "n-best read-out of a decoded batch", -1, methods};

PyMODINIT_FUNC PyInit__readout(void) { return PyModule_Create(&moddef); }

/* Host-side helper (CPython extension, no GPU code): the reference's DecodeResult carries the
 * n-best of a prefix beam search as plain Python lists (wenet/models/transformer/search.py:30-61:
 * nbest = list of token tuples, nbest_scores = list of floats, nbest_times = list of lists).
 * wn_ctc_prefix_beam_search returns them as padded arrays of the whole batch; this module turns
 * those arrays into the three list families of EVERY utterance in one pass over the used
 * elements only (one PyLong per token / time stamp, no intermediate padded lists).
 *
 *   build(n_hyps[B] i32, hyp_lens[B,beam] i32, hyp_tlens[B,beam] i32,
 *         hyp_tokens[B,beam,max_len] i32, hyp_times[B,beam,max_len] i32,
 *         hyp_scores[B,beam] f64, B, beam, max_len)
 *     -> list of B tuples (nbest, nbest_scores, nbest_times)
 *
 * Arguments are C-contiguous buffers (numpy arrays); shapes are passed explicitly and checked
 * against the buffer sizes.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static int get_buf(PyObject* o, Py_buffer* v, Py_ssize_t need_bytes, const char* name) {
  if (PyObject_GetBuffer(o, v, PyBUF_C_CONTIGUOUS) != 0) return -1;
  if (v->len < need_bytes) {
    PyErr_Format(PyExc_ValueError, "nbest_lists.build: %s holds %zd bytes, %zd needed", name,
                 v->len, need_bytes);
    PyBuffer_Release(v);
    return -1;
  }
  return 0;
}

/* small non-negative ints dominate (token ids < vocabulary, frame indices < T'): a per-call
 * table of the PyLongs already made avoids one allocation per repeated value */
#define MEMO 8192

static PyObject* build(PyObject* self, PyObject* args) {
  PyObject *o_n, *o_len, *o_tlen, *o_tok, *o_tim, *o_sc;
  Py_ssize_t B, beam, max_len;
  if (!PyArg_ParseTuple(args, "OOOOOOnnn", &o_n, &o_len, &o_tlen, &o_tok, &o_tim, &o_sc, &B,
                        &beam, &max_len))
    return NULL;
  if (B < 0 || beam < 1 || max_len < 1) {
    PyErr_SetString(PyExc_ValueError, "nbest_lists.build: bad shape");
    return NULL;
  }
  Py_buffer vn, vlen, vtlen, vtok, vtim, vsc;
  if (get_buf(o_n, &vn, B * 4, "n_hyps")) return NULL;
  if (get_buf(o_len, &vlen, B * beam * 4, "hyp_lens")) { PyBuffer_Release(&vn); return NULL; }
  if (get_buf(o_tlen, &vtlen, B * beam * 4, "hyp_tlens")) {
    PyBuffer_Release(&vn); PyBuffer_Release(&vlen); return NULL;
  }
  if (get_buf(o_tok, &vtok, B * beam * max_len * 4, "hyp_tokens")) {
    PyBuffer_Release(&vn); PyBuffer_Release(&vlen); PyBuffer_Release(&vtlen); return NULL;
  }
  if (get_buf(o_tim, &vtim, B * beam * max_len * 4, "hyp_times")) {
    PyBuffer_Release(&vn); PyBuffer_Release(&vlen); PyBuffer_Release(&vtlen);
    PyBuffer_Release(&vtok); return NULL;
  }
  if (get_buf(o_sc, &vsc, B * beam * 8, "hyp_scores")) {
    PyBuffer_Release(&vn); PyBuffer_Release(&vlen); PyBuffer_Release(&vtlen);
    PyBuffer_Release(&vtok); PyBuffer_Release(&vtim); return NULL;
  }
  const int32_t* n_hyps = (const int32_t*)vn.buf;
  const int32_t* lens = (const int32_t*)vlen.buf;
  const int32_t* tlens = (const int32_t*)vtlen.buf;
  const int32_t* tok = (const int32_t*)vtok.buf;
  const int32_t* tim = (const int32_t*)vtim.buf;
  const double* sc = (const double*)vsc.buf;

  PyObject** memo = (PyObject**)PyMem_Calloc(MEMO, sizeof(PyObject*));
  PyObject* out = memo ? PyList_New(B) : NULL;
  int ok = out != NULL;
#define LONG_OF(dst, v)                                                  \
  do {                                                                   \
    int32_t v_ = (v);                                                    \
    if (v_ >= 0 && v_ < MEMO) {                                          \
      if (!memo[v_]) memo[v_] = PyLong_FromLong(v_);                     \
      (dst) = memo[v_];                                                  \
      Py_XINCREF(dst);                                                   \
    } else {                                                             \
      (dst) = PyLong_FromLong(v_);                                       \
    }                                                                    \
  } while (0)
  for (Py_ssize_t b = 0; ok && b < B; ++b) {
    Py_ssize_t n = n_hyps[b];
    if (n < 0) n = 0;
    if (n > beam) n = beam;
    PyObject* nb = PyList_New(n);
    PyObject* ns = PyList_New(n);
    PyObject* nt = PyList_New(n);
    PyObject* rec = (nb && ns && nt) ? PyTuple_New(3) : NULL;
    if (!rec) { Py_XDECREF(nb); Py_XDECREF(ns); Py_XDECREF(nt); ok = 0; break; }
    PyTuple_SET_ITEM(rec, 0, nb);
    PyTuple_SET_ITEM(rec, 1, ns);
    PyTuple_SET_ITEM(rec, 2, nt);
    PyList_SET_ITEM(out, b, rec);
    for (Py_ssize_t i = 0; ok && i < n; ++i) {
      Py_ssize_t L = lens[b * beam + i], Lt = tlens[b * beam + i];
      if (L < 0) L = 0;
      if (L > max_len) L = max_len;
      if (Lt < 0) Lt = 0;
      if (Lt > max_len) Lt = max_len;
      const int32_t* tp = tok + (b * beam + i) * max_len;
      const int32_t* mp = tim + (b * beam + i) * max_len;
      PyObject* t = PyTuple_New(L);
      PyObject* m = PyList_New(Lt);
      PyObject* s = PyFloat_FromDouble(sc[b * beam + i]);
      if (!t || !m || !s) { Py_XDECREF(t); Py_XDECREF(m); Py_XDECREF(s); ok = 0; break; }
      PyList_SET_ITEM(nb, i, t);
      PyList_SET_ITEM(nt, i, m);
      PyList_SET_ITEM(ns, i, s);
      for (Py_ssize_t k = 0; k < L; ++k) {
        PyObject* v;
        LONG_OF(v, tp[k]);
        if (!v) { ok = 0; break; }
        PyTuple_SET_ITEM(t, k, v);
      }
      for (Py_ssize_t k = 0; ok && k < Lt; ++k) {
        PyObject* v;
        LONG_OF(v, mp[k]);
        if (!v) { ok = 0; break; }
        PyList_SET_ITEM(m, k, v);
      }
    }
  }
#undef LONG_OF
  if (memo) {
    for (int i = 0; i < MEMO; ++i) Py_XDECREF(memo[i]);
    PyMem_Free(memo);
  }
  PyBuffer_Release(&vn); PyBuffer_Release(&vlen); PyBuffer_Release(&vtlen);
  PyBuffer_Release(&vtok); PyBuffer_Release(&vtim); PyBuffer_Release(&vsc);
  if (!ok) {
    Py_XDECREF(out);
    if (!PyErr_Occurred()) PyErr_NoMemory();
    return NULL;
  }
  return out;
}

static PyMethodDef methods[] = {
    {"build", build, METH_VARARGS,
     "build(n_hyps, hyp_lens, hyp_tlens, hyp_tokens, hyp_times, hyp_scores, B, beam, max_len)"
     " -> [(nbest, nbest_scores, nbest_times)] * B"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef mod = {PyModuleDef_HEAD_INIT, "_nbest_lists",
                                 "n-best arrays of a batch -> the reference's list fields", -1,
                                 methods};

PyMODINIT_FUNC PyInit__nbest_lists(void) { return PyModule_Create(&mod); }

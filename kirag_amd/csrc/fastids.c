/* _fastids: the id-string pass of Indexer.search_knn as one C loop.
 *
 * The reference maps every hit to str(id) in a Python list comprehension (retriever/index.py:49): 102 400 str() calls per 1024-query x top-100 block.  Round 4 moved the
 * formatting into the library (kr_format_ids: one ASCII buffer) but still paid bytes -> str decode, str.split and 1024 list slices in the interpreter - together as
 * long as the GPU's whole search of the block on a slow host.  This extension builds the List[List[str]] directly: decimal digits into a
 * stack buffer, PyUnicode_New + memcpy (ASCII), PyList_SET_ITEM.  Host-side plumbing only: no arithmetic of the path lives here; when the extension is missing
 * (no Python.h at build time, another interpreter) kirag_amd.retriever.flat_index falls back to kr_format_ids + split with identical results.
 *
 * Built by kirag_amd/csrc/Makefile (gcc, CPython C API only - no numpy headers: arrays arrive through the buffer protocol). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

static inline int fmt_i64(int64_t v, char* out /* >= 20 bytes */) {
    char tmp[20];
    int n = 0, neg = v < 0;
    uint64_t u = neg ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    do { tmp[n++] = (char)('0' + (u % 10)); u /= 10; } while (u);
    int len = 0;
    if (neg) out[len++] = '-';
    while (n) out[len++] = tmp[--n];
    return len;
}

/* ids_to_str_rows(ids: buffer of int64 (C-contiguous), nq: int, k: int) -> list of nq lists of k str */
static PyObject* ids_to_str_rows(PyObject* self, PyObject* args) {
    Py_buffer view;
    Py_ssize_t nq, k;
    (void)self;
    if (!PyArg_ParseTuple(args, "y*nn", &view, &nq, &k)) return NULL;
    if (nq < 0 || k < 0 || (k > 0 && nq > PY_SSIZE_T_MAX / k) || view.len != nq * k * (Py_ssize_t)sizeof(int64_t)) {
        PyBuffer_Release(&view);
        PyErr_SetString(PyExc_ValueError, "ids_to_str_rows: the buffer must hold exactly nq * k int64 values");
        return NULL;
    }
    const int64_t* ids = (const int64_t*)view.buf;
    PyObject* outer = PyList_New(nq);
    if (!outer) { PyBuffer_Release(&view); return NULL; }
    char buf[24];
    for (Py_ssize_t q = 0; q < nq; ++q) {
        PyObject* row = PyList_New(k);
        if (!row) goto fail;
        PyList_SET_ITEM(outer, q, row);
        for (Py_ssize_t j = 0; j < k; ++j) {
            const int len = fmt_i64(ids[q * k + j], buf);
            PyObject* s = PyUnicode_New(len, 127);
            if (!s) goto fail;
            memcpy(PyUnicode_DATA(s), buf, (size_t)len);
            PyList_SET_ITEM(row, j, s);
        }
    }
    PyBuffer_Release(&view);
    return outer;
fail:
    PyBuffer_Release(&view);
    Py_DECREF(outer);       /* rows already stored are released with it; unset slots are NULL, which list dealloc accepts */
    return NULL;
}

static PyMethodDef methods[] = {
    {"ids_to_str_rows", ids_to_str_rows, METH_VARARGS, "int64 ids [nq*k] -> List[List[str]] of their decimal strings"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_fastids", "id-string pass of Indexer.search_knn (retriever/index.py:49) in C", -1, methods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__fastids(void) { return PyModule_Create(&moddef); }

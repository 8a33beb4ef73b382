/* lsfast.c - a 60-line CPython binding of ONE entry point, ls_search (include/leansearch.h), for the
 * reference's hot call `index.search(x, k)` (search/engine.py:250). ctypes costs ~2.5 us per call here
 * (argument conversion of seven arguments + three pointer look-ups) against a 60 us search; this module
 * takes the three arrays through the buffer protocol and calls the library's function pointer with the GIL
 * released (as ctypes does: concurrent callers are combined inside the library). Nothing is computed
 * here; without this module lean_explore_amd.index falls back to the ctypes binding of the same symbol. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef int (*ls_search_fn)(void*, const float*, int64_t, int32_t, uint32_t, float*, int64_t*);

/* search(fn_addr, handle, x, nq, k, flags, D, I) -> rc */
static PyObject* lsfast_search(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
    (void)self;
    if (nargs != 8) {
        PyErr_SetString(PyExc_TypeError, "search(fn_addr, handle, x, nq, k, flags, D, I)");
        return NULL;
    }
    ls_search_fn fn = (ls_search_fn)PyLong_AsVoidPtr(args[0]);
    void* handle = PyLong_AsVoidPtr(args[1]);
    const long long nq = PyLong_AsLongLong(args[3]);
    const long k = PyLong_AsLong(args[4]);
    const unsigned long flags = PyLong_AsUnsignedLong(args[5]);
    if (PyErr_Occurred()) return NULL;
    Py_buffer x, D, I;
    if (PyObject_GetBuffer(args[2], &x, PyBUF_SIMPLE) < 0) return NULL;
    if (PyObject_GetBuffer(args[6], &D, PyBUF_WRITABLE) < 0) {
        PyBuffer_Release(&x);
        return NULL;
    }
    if (PyObject_GetBuffer(args[7], &I, PyBUF_WRITABLE) < 0) {
        PyBuffer_Release(&x);
        PyBuffer_Release(&D);
        return NULL;
    }
    int rc = -1;
    /* the caller (index.py) has checked the shapes; the sizes are re-checked here because a short
     * buffer would be written past its end */
    if (nq >= 0 && k > 0 && D.len >= (Py_ssize_t)(nq * k * 4) && I.len >= (Py_ssize_t)(nq * k * 8) && fn && handle) {
        Py_BEGIN_ALLOW_THREADS
        rc = fn(handle, nq ? (const float*)x.buf : NULL, nq, (int32_t)k, (uint32_t)flags, nq ? (float*)D.buf : NULL,
                nq ? (int64_t*)I.buf : NULL);
        Py_END_ALLOW_THREADS
    } else {
        PyErr_SetString(PyExc_ValueError, "lsfast.search: output buffers too small or null handle");
    }
    PyBuffer_Release(&x);
    PyBuffer_Release(&D);
    PyBuffer_Release(&I);
    if (PyErr_Occurred()) return NULL;
    return PyLong_FromLong(rc);
}

static PyMethodDef methods[] = {{"search", (PyCFunction)(void (*)(void))lsfast_search, METH_FASTCALL,
                                 "search(fn_addr, handle, x, nq, k, flags, D, I) -> rc of ls_search"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_lsfast", NULL, -1, methods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__lsfast(void) { return PyModule_Create(&moddef); }

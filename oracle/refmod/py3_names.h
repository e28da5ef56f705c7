/* oracle/refmod/py3_names.h -- TEST INFRASTRUCTURE, build container only.
 *
 * Force-included (gcc -include) in front of the reference's UNMODIFIED reveal.c / interface.c so that the module they
 * define can be imported by this image's CPython 3 and its real aligner() (reveal.c:731-1338) can run with Python 3
 * callbacks.  The reference is a Python-2 extension; four spellings differ:
 *   Py_InitModule3            interface.c:902,925   (module creation)
 *   PyInt_AS_LONG             reveal.c:923          (the only place a match position is read back from Python)
 *   PyString_Check            interface.c:29
 *   PyArg_ParseTuple "s#"     interface.c:58        (needs an int length without PY_SSIZE_T_CLEAN)
 *   PyObject_HEAD_INIT(NULL) 0, "reveal", ...       interface.c:842-844 (Py2 layout of the static type's head)
 * These are names of the CPython API of another major version mapped onto this image's CPython; no header, library
 * or generated file of the reference is replaced, nothing of its algorithmic code is touched.  The primary pin of
 * the oracle stays oracle/_ref/libreveal_ref[64].so (the same sources with NO shim, plain-C entry points only);
 * this module is the second checker VERDICT r1 asks for: the one way to execute aligner() itself.
 */
#include <Python.h>
#include <ctype.h>
#include <assert.h>
PyObject *Py_InitModule3(const char *, PyMethodDef *, const char *);
long PyInt_AS_LONG(PyObject *);
int PyString_Check(PyObject *);
int refmod_ParseTuple(PyObject *, const char *, ...);
#define PyArg_ParseTuple refmod_ParseTuple
#undef PyObject_HEAD_INIT
#define PyObject_HEAD_INIT(type) 1, type,

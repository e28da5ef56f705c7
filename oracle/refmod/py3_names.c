/* oracle/refmod/py3_names.c -- see py3_names.h.  TEST INFRASTRUCTURE, build container only. */
#include <Python.h>
#include <string.h>
#include <stdarg.h>

static PyObject *g_mod;
static struct PyModuleDef g_def = {PyModuleDef_HEAD_INIT, "reveallib", "reference reveallib (Py2 extension) under CPython 3", -1, NULL};

PyObject *Py_InitModule3(const char *name, PyMethodDef *methods, const char *doc) {
    g_def.m_name = name; g_def.m_doc = doc; g_def.m_methods = methods;
    g_mod = PyModule_Create(&g_def);
    return g_mod;
}
long PyInt_AS_LONG(PyObject *o) { return PyLong_AsLong(o); }
int PyString_Check(PyObject *o) { return PyUnicode_Check(o); }

/* interface.c:58 parses "s#" into (char *, int); everything else goes to the real parser */
int refmod_ParseTuple(PyObject *args, const char *fmt, ...) {
    va_list va;
    int r;
    va_start(va, fmt);
    if (strcmp(fmt, "s#") == 0) {
        char **s = va_arg(va, char **);
        int *l = va_arg(va, int *);
        PyObject *o = PyTuple_GetItem(args, 0);
        Py_ssize_t n = 0;
        const char *p = NULL;
        if (o) {
            if (PyBytes_Check(o)) { p = PyBytes_AsString(o); n = PyBytes_Size(o); }
            else p = PyUnicode_AsUTF8AndSize(o, &n);
        }
        if (p) { *s = (char *)p; *l = (int)n; r = 1; } else r = 0;
    } else
        r = PyArg_VaParse(args, fmt, va);
    va_end(va);
    return r;
}

#ifdef SA64
void initreveallib64(void);
PyMODINIT_FUNC PyInit_reveallib64(void) { initreveallib64(); return g_mod; }
#else
void initreveallib(void);
PyMODINIT_FUNC PyInit_reveallib(void) { initreveallib(); return g_mod; }
#endif

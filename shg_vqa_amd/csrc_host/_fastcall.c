/* Low-overhead trampoline from Python to the C ABI of libshgvqa.so.
 *
 * A training step is ~1 200 kernel launches issued from Python; ctypes spends ~3-4 us per call converting
 * 15-30 arguments, which is a tenth of the step once the GPU side is fast.  This module converts the arguments
 * with the CPython C API (~0.5 us) and calls the entry point through one fixed prototype.
 *
 * Calling convention (x86-64 System V, the only host this library targets): integer-class arguments take the next
 * integer register / stack slot and float arguments the next xmm register, independently of how they interleave in
 * the callee's signature, and surplus arguments are ignored by the callee.  Every entry point of include/shg_vqa.h
 * takes only pointers, 32/64-bit integers and floats (<= 8 floats), so all of them can be called as
 *     int f(long x 30, float x 8)
 * with the integer-class arguments in order in the longs and the floats in order in the floats.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#if !defined(__x86_64__) || defined(_WIN32)
#error "_fastcall relies on the x86-64 System V calling convention"
#endif

#define MAXI 30
#define MAXF 8
typedef int (*entry_t)(long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long, long,
                       long, long, long, long, long, long, long, long, long, long, long, long, float, float, float, float, float,
                       float, float, float);

/* call(fn_address, signature_bytes, *args) -> int
 * signature: one char per argument: 'p' pointer / 'l' int64 / 'i' int32 / 'u' uint64 (integer class), 'f' float */
static PyObject* fast_call(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs < 2) {
        PyErr_SetString(PyExc_TypeError, "call(fn, sig, *args)");
        return NULL;
    }
    entry_t fn = (entry_t)PyLong_AsVoidPtr(args[0]);
    if (!fn && PyErr_Occurred()) return NULL;
    char* sig;
    Py_ssize_t nsig;
    if (PyBytes_AsStringAndSize(args[1], &sig, &nsig) < 0) return NULL;
    if (nsig != nargs - 2) {
        PyErr_Format(PyExc_TypeError, "expected %zd arguments, got %zd", nsig, nargs - 2);
        return NULL;
    }
    long iv[MAXI] = {0};
    float fv[MAXF] = {0};
    int ni = 0, nf = 0;
    for (Py_ssize_t k = 0; k < nsig; ++k) {
        PyObject* a = args[2 + k];
        if (sig[k] == 'f') {
            if (nf >= MAXF) { PyErr_SetString(PyExc_TypeError, "too many float arguments"); return NULL; }
            double d = PyFloat_AsDouble(a);
            if (d == -1.0 && PyErr_Occurred()) return NULL;
            fv[nf++] = (float)d;
        } else {
            if (ni >= MAXI) { PyErr_SetString(PyExc_TypeError, "too many integer arguments"); return NULL; }
            long v;
            if (a == Py_None) v = 0;
            else if (sig[k] == 'u') {
                v = (long)PyLong_AsUnsignedLongLongMask(a);
                if (v == -1 && PyErr_Occurred()) return NULL;
            } else {
                v = PyLong_AsLong(a);
                if (v == -1 && PyErr_Occurred()) {
                    /* addresses above 2^63 do not occur in user space; report the conversion error */
                    return NULL;
                }
            }
            iv[ni++] = v;
        }
    }
    int rc = fn(iv[0], iv[1], iv[2], iv[3], iv[4], iv[5], iv[6], iv[7], iv[8], iv[9], iv[10], iv[11], iv[12], iv[13], iv[14],
                iv[15], iv[16], iv[17], iv[18], iv[19], iv[20], iv[21], iv[22], iv[23], iv[24], iv[25], iv[26], iv[27], iv[28],
                iv[29], fv[0], fv[1], fv[2], fv[3], fv[4], fv[5], fv[6], fv[7]);
    return PyLong_FromLong(rc);
}

static PyMethodDef methods[] = {
    {"call", (PyCFunction)(void (*)(void))fast_call, METH_FASTCALL, "call(fn_address, signature, *args) -> int"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_fastcall", "low-overhead calls into libshgvqa.so", -1, methods};

PyMODINIT_FUNC PyInit__fastcall(void) { return PyModule_Create(&module); }

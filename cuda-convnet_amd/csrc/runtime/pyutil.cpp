#include "pyutil.h"

#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

#include <stdexcept>

#include "devmat.h"

namespace rt {

static void* init_numpy_impl() {
    import_array();
    return nullptr;
}
void initNumpy() { init_numpy_impl(); }

HostMat::HostMat(PyObject* o) {
    if (!o || !PyArray_Check(o)) throw std::runtime_error("expected a numpy.ndarray");
    PyArrayObject* a = reinterpret_cast<PyArrayObject*>(o);
    if (PyArray_NDIM(a) != 2) throw std::runtime_error(fmt("expected a 2-D array, got %d-D", PyArray_NDIM(a)));
    const bool f32 = PyArray_TYPE(a) == NPY_FLOAT32;
    if (f32 && PyArray_IS_C_CONTIGUOUS(a)) {
        Py_INCREF(o);
        _obj = o;
    } else if (f32 && PyArray_IS_F_CONTIGUOUS(a)) {
        Py_INCREF(o);
        _obj = o;
        _fortran = true;
    } else {
        _obj = PyArray_FROM_OTF(o, NPY_FLOAT32, NPY_ARRAY_C_CONTIGUOUS | NPY_ARRAY_ALIGNED | NPY_ARRAY_FORCECAST);
        if (!_obj) throw std::runtime_error("unable to convert array to contiguous float32");
    }
    PyArrayObject* b = reinterpret_cast<PyArrayObject*>(_obj);
    _rows = (int)PyArray_DIM(b, 0);
    _cols = (int)PyArray_DIM(b, 1);
    _data = static_cast<float*>(PyArray_DATA(b));
}

void HostMat::release() {
    Py_XDECREF(_obj);
    _obj = nullptr;
    _data = nullptr;
}

void HostMat::toRowMajor(float* dst) const {
    if (!_fortran) {
        memcpy(dst, _data, numel() * sizeof(float));
        return;
    }
    for (int r = 0; r < _rows; ++r)
        for (int c = 0; c < _cols; ++c) dst[(size_t)r * _cols + c] = _data[(size_t)c * _rows + r];
}

void HostMat::fromRowMajor(const float* src) {
    if (!_fortran) {
        memcpy(_data, src, numel() * sizeof(float));
        return;
    }
    for (int r = 0; r < _rows; ++r)
        for (int c = 0; c < _cols; ++c) _data[(size_t)c * _rows + r] = src[(size_t)r * _cols + c];
}

static PyObject* item(PyObject* dict, const char* key, bool required = true) {
    PyObject* o = PyDict_GetItemString(dict, key);
    if (!o && required) throw std::runtime_error(fmt("layer dictionary has no key '%s'", key));
    return o;
}

bool pyDictHas(PyObject* dict, const char* key) { return PyDict_GetItemString(dict, key) != nullptr; }

int pyDictGetInt(PyObject* dict, const char* key) {
    PyObject* o = item(dict, key);
    long v = PyLong_AsLong(o);
    if (v == -1 && PyErr_Occurred()) {
        PyErr_Clear();
        double d = PyFloat_AsDouble(o);
        if (d == -1.0 && PyErr_Occurred()) {
            PyErr_Clear();
            throw std::runtime_error(fmt("key '%s' is not an integer", key));
        }
        v = (long)d;
    }
    return (int)v;
}

float pyDictGetFloat(PyObject* dict, const char* key) {
    double d = PyFloat_AsDouble(item(dict, key));
    if (d == -1.0 && PyErr_Occurred()) {
        PyErr_Clear();
        throw std::runtime_error(fmt("key '%s' is not a number", key));
    }
    return (float)d;
}

std::string pyDictGetString(PyObject* dict, const char* key) {
    PyObject* o = item(dict, key);
    const char* s = PyUnicode_Check(o) ? PyUnicode_AsUTF8(o) : (PyBytes_Check(o) ? PyBytes_AsString(o) : nullptr);
    if (!s) {
        PyErr_Clear();
        throw std::runtime_error(fmt("key '%s' is not a string", key));
    }
    return std::string(s);
}

static PyObject* as_seq(PyObject* o, const char* what) {
    PyObject* seq = PySequence_Fast(o, what);
    if (!seq) {
        PyErr_Clear();
        throw std::runtime_error(what);
    }
    return seq;
}

intv* getIntV(PyObject* list) {
    PyObject* seq = as_seq(list, "expected a list of ints");
    intv* v = new intv();
    for (Py_ssize_t i = 0; i < PySequence_Fast_GET_SIZE(seq); ++i) {
        PyObject* e = PySequence_Fast_GET_ITEM(seq, i);
        long x = PyLong_AsLong(e);
        if (x == -1 && PyErr_Occurred()) {
            PyErr_Clear();
            x = (long)PyFloat_AsDouble(e);
            PyErr_Clear();
        }
        v->push_back((int)x);
    }
    Py_DECREF(seq);
    return v;
}

intv* pyDictGetIntV(PyObject* dict, const char* key) {
    PyObject* o = item(dict, key, false);
    return o ? getIntV(o) : nullptr;
}

floatv* pyDictGetFloatV(PyObject* dict, const char* key) {
    PyObject* o = item(dict, key, false);
    if (!o) return nullptr;
    PyObject* seq = as_seq(o, "expected a list of floats");
    floatv* v = new floatv();
    for (Py_ssize_t i = 0; i < PySequence_Fast_GET_SIZE(seq); ++i) v->push_back((float)PyFloat_AsDouble(PySequence_Fast_GET_ITEM(seq, i)));
    Py_DECREF(seq);
    if (PyErr_Occurred()) {
        PyErr_Clear();
        delete v;
        throw std::runtime_error(fmt("key '%s' is not a list of numbers", key));
    }
    return v;
}

HostMat* pyDictGetMatrix(PyObject* dict, const char* key) { return new HostMat(item(dict, key)); }

HostMatV* getMatrixV(PyObject* list) {
    PyObject* seq = as_seq(list, "expected a list of ndarrays");
    HostMatV* v = new HostMatV();
    try {
        for (Py_ssize_t i = 0; i < PySequence_Fast_GET_SIZE(seq); ++i) v->push_back(new HostMat(PySequence_Fast_GET_ITEM(seq, i)));
    } catch (...) {
        for (HostMat* m : *v) { m->release(); delete m; }
        delete v;
        Py_DECREF(seq);
        throw;
    }
    Py_DECREF(seq);
    return v;
}

HostMatV* pyDictGetMatrixV(PyObject* dict, const char* key) { return getMatrixV(item(dict, key)); }

}  // namespace rt
